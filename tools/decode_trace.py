#!/usr/bin/env python
"""Round 6: the launch sequence of ONE lone decode forward (kernel names, durations, gaps) from a rocprofv3 kernel trace.
  run:      rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/decode_trace.py run cfg2|cfg3|cfg5 [images]
  analyse:  python tools/decode_trace.py show <kernel_trace.csv> [timestep]
`show` prints the dispatches of the LAST forward between two pack_input launches: per timestep (split at the first controller-CNN
launch) every kernel with its duration and the idle gap before it, then the sums."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cfg, images):
  sys.path.insert(0, ROOT)
  sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd'))
  import torch
  import bench
  import full_model
  box = cfg.endswith('box')
  cfg = cfg[:-3] if box else cfg
  if cfg == 'cfg2':
    arch, H, W, T, B = 'cvppp', 512, 512, 16, 8
  else:
    c = bench.OTHER_CONFIGS[cfg]
    arch, H, W, T, B = c['arch'], c['H'], c['W'], c['T'], c['B']
  B = images or B
  opt = bench.make_opt(arch, H, W, T)
  if box:
    import box_model
    m = box_model.get_model(opt)
  else:
    m = full_model.get_model(opt, is_training=False)
  bench.seed_weights(m, 1234)
  g = torch.Generator().manual_seed(1234)
  x = torch.rand((B, H, W, 3), generator=g).cuda()
  kw = {}
  if opt['add_d_out']:
    kw['d_in'] = torch.nn.functional.one_hot(torch.randint(0, 8, (B, H, W), generator=g), 8).float().cuda()
    kw['y_in'] = torch.softmax(torch.randn((B, H, W, opt['num_semantic_classes']), generator=g), dim=-1).cuda()
  if box:
    yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    y_gt = torch.zeros((B, T, H, W))
    for t in range(min(T, 6)):
      y_gt[:, t] = (((yy - (20 + 15 * t) % H) ** 2 / 400.0 + (xx - (30 + 60 * t) % W) ** 2 / 900.0) <= 1).float()
    kw['y_gt'] = y_gt.cuda()
  for _ in range(4):
    m.engine.forward(x, **kw)
    torch.cuda.synchronize()


def short(n):
  n = n.replace('void ', '')
  m = re.search(r'ra::\w+::\w+(<[^>]*>)?', n)
  return m.group(0) if m else n[:70]


def show(path, which=None):
  import csv
  rows = list(csv.DictReader(open(path)))
  rows.sort(key=lambda r: int(r['Start_Timestamp']))
  packs = [i for i, r in enumerate(rows) if 'pack_input_kernel' in r['Kernel_Name']]
  a = packs[-1]
  seq = rows[a:]
  t_prev_end = int(seq[0]['Start_Timestamp'])
  steps, cur = [], []
  first_enc = None
  for r in seq:
    s = short(r['Kernel_Name'])
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    item = (s, (en - st) / 1e3, (st - t_prev_end) / 1e3)
    t_prev_end = en
    if first_enc is None and ('cpair' in s or 'conv3x3' in s or 'csplit' in s or 'wino' in s):
      first_enc = s
    if s == first_enc and cur and any(('paste' in c[0] or 'canvas_max' in c[0]) for c in cur):
      steps.append(cur)
      cur = []
    cur.append(item)
  steps.append(cur)
  print('forward: %d dispatches, %d timesteps, %.1f us from first start to last end' % (
      len(seq), len(steps), (int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e3))
  k = int(which) if which is not None else min(2, len(steps) - 1)
  print('timestep %d:' % k)
  for s, d, gap in steps[k]:
    print('  %8.2f us  (gap %6.2f)  %s' % (d, gap, s))
  for k, st in enumerate(steps):
    print('step %2d: %3d launches, kernels %8.1f us, gaps %7.1f us' % (k, len(st), sum(d for _, d, _ in st), sum(g for _, _, g in st)))


if __name__ == '__main__':
  if sys.argv[1] == 'run':
    run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
  else:
    show(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
