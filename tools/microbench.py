#!/usr/bin/env python
"""Per-kernel timing on the GPU box (HIP events on the launch stream, many back-to-back
launches).  Usage: python tools/microbench.py [--batch 8] [--size 512] [names...]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd'))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import full_model
import ra_ops as ops


def timeit(fn, reps=20, inner=8):
  """In-graph timing: `inner` copies captured in one HIP graph (amortises the ~10 us replay cost)."""
  fn()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(inner):
      fn()
  for _ in range(3):
    g.replay()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  e0.record()
  for _ in range(reps):
    g.replay()
  e1.record()
  torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / (reps * inner)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--batch', type=int, default=8)
  ap.add_argument('--size', type=int, default=512)
  ap.add_argument('names', nargs='*')
  args = ap.parse_args()
  B, S, T = args.batch, args.size, 2
  opt = bench.make_opt('cvppp', S, S, T)
  m = full_model.get_model(opt, is_training=False)
  bench.seed_weights(m, 1)
  eng = m.engine
  eng.use_graph = False
  eng.nsub = 1
  x = torch.rand((B, S, S, 3)).cuda()
  eng.forward(x)
  torch.cuda.synchronize()
  d, b, Wt = eng.d, eng.subs[0], eng.W
  H = W = S
  res = {}

  def want(n):
    return not args.names or any(n.startswith(a) for a in args.names)

  src = b['img']
  tot_f, per_f = bench.encoder_flops_per_image(d)
  for i, (wp, sc, sh, cout, pool) in enumerate(Wt['ccnn']):
    if want('conv_L'):
      s_ = src
      us = timeit(lambda: ops.conv3x3(s_, wp, sc[0], sh[0], cout, relu=True, pool=pool, out=b['ccnn'][i]))
      res['conv_L%d' % i] = (us, per_f[i] * B / us / 1e6)
    src = b['ccnn'][i]
  if want('pair'):
    for step in eng.plan['ccnn']:
      first = step[1]
      s0 = b['img'] if first == 0 else b['ccnn'][first - 1]
      fl = sum(per_f[i] for i in step[1:])
      us = timeit(lambda: eng._run_cnn([step], Wt['ccnn'], s0, b['ccnn'], 0, 'x'))
      res['pair_' + '+'.join(str(i) for i in step[1:])] = (us, fl * B / us / 1e6)
  if want('controller'):
    res['controller'] = (timeit(lambda: ops.controller(eng.desc, src, Wt['ctrl'], b['h_last'][0],
                                                       b['ctrl_out'][0], b['gmaps'][0], b['attn'][0])), 0)
  if want('extract'):
    us = timeit(lambda: ops.extract_direct(b['img'], 0, b['attn'][0], 48, 48, d['C0p'], True, b['x_patch'][0],
                                           canvas=b['canvas'], canvas_chan=d['D']))
    res['extract'] = (us, 0)
  if want('paste'):
    pt = b['y_out_patch'][0]
    us = timeit(lambda: ops.paste_direct(pt, 0, b['attn'][0], -5.0, False, b['y_out'].data_ptr(), T * H * W, H, W,
                                         canvas=b['canvas'], flags=ops.PASTE_Y_PREFILLED | ops.PASTE_CANVAS_FLOORED))
    res['paste'] = (us, (S * S * 12.0 * B) / us / 1e3)
  if want('patch'):
    s2 = b['x_patch'][0]
    for i, (wp, sc, sh, cout, pool) in enumerate(Wt['acnn']):
      s_ = s2
      res['acnn_L%d' % i] = (timeit(lambda: ops.conv3x3(s_, wp, sc[0], sh[0], cout, relu=True, pool=pool, out=b['acnn'][i])), 0)
      s2 = b['acnn'][i]
    for i, (wp, sc, sh, cout, unpool, sidx) in enumerate(Wt['adcnn']):
      out = b['y_out_patch'][0] if b['adcnn'][i] is None else b['adcnn'][i]
      s_ = s2
      res['adcnn_L%d' % i] = (timeit(lambda: ops.conv3x3(s_, wp, sc[0], sh[0], cout, relu=True, pool=1, upsample=(unpool == 2), out=out)), 0)
      s2 = out
  if want('acnn'):
    def chain():
      s2 = b['x_patch'][0]
      for i, (wp, sc, sh, cout, pool) in enumerate(Wt['acnn']):
        ops.conv3x3(s2, wp, sc[0], sh[0], cout, relu=True, pool=pool, out=b['acnn'][i])
        s2 = b['acnn'][i]
      for i, (wp, sc, sh, cout, unpool, sidx) in enumerate(Wt['adcnn']):
        out = b['y_out_patch'][0] if b['adcnn'][i] is None else b['adcnn'][i]
        ops.conv3x3(s2, wp, sc[0], sh[0], cout, relu=True, pool=1, upsample=(unpool == 2), out=out)
        s2 = out
    res['acnn+adcnn(13 launches)'] = (timeit(chain), 0)
  if want('score'):
    core = b['acnn'][-1]
    res['score'] = (timeit(lambda: ops.dense(b['h_last'][0], Wt['smlp_w'], Wt['smlp_b'], 'sigmoid',
                                             b['s_out'].data_ptr(), T, x1=core.view(B, -1))), 0)
  for k, (us, rate) in res.items():
    print('%-28s %9.2f us   %s' % (k, us, ('%.1f TF/s|GB/s' % rate) if rate else ''))


if __name__ == '__main__':
  main()
