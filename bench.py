#!/usr/bin/env python
"""Benchmark of the recurrent-attention decode loop on MI355X.

One "step" = one eval-mode full_model forward (all T instance timesteps) of a batch of B
synthetic CVPPP-shaped images per GPU: the workload BASELINE.json's metric is quoted on
(configs[1]: CVPPP arch, 512x512, T=16, B=8).  Inputs and weights are resident in HBM before
the timed region.  Multi-GPU: one process per GPU, every rank decodes its own shard of B
images (the path is embarrassingly parallel over the batch; no data-path collective), so
scaling is weak.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel group — the controller
CNN's conv3x3 f32-MFMA launches (1.585 GFLOP per image-timestep at cfg2) —
timed with HIP events on the launch stream; `roofline_attn` the HBM-bound attention resample
(extract + paste, 7.0 MiB algorithmic per image-timestep); `cpu_baseline` the PyTorch-CPU
restatement (float32, all host cores) timed over a bounded sample (a reported baseline, not the
target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd'))

import numpy as np
import torch
import ra_native  # before the first HIP call: it sets the runtime's hardware-queue count (GPU_MAX_HW_QUEUES)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, exact f32
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)


def make_opt(arch, H, W, T):
  """model_opt of the reference's run scripts (run_cvppp.sh:37-72 / run_kitti.sh:68-111 via
  full_model_train.py:581-658)."""
  opt = dict(
      inp_height=H, inp_width=W, inp_depth=3, padding=16, filter_height=48, filter_width=48,
      timespan=T, ctrl_rnn_hid_dim=256, num_ctrl_mlp_layers=1, ctrl_mlp_dim=256,
      mlp_dropout=None, weight_decay=5e-5, use_bn=True, attn_box_padding_ratio=0.2,
      use_knob=False, squash_ctrl_params=False, fixed_order=False, fixed_gamma=True,
      fixed_var=False, dynamic_var=False, num_ctrl_rnn_iter=5, num_glimpse_mlp_layers=2,
      stop_canvas_grad=True, use_iou_box=False, add_skip_conn=False, disable_overwrite=False,
      add_d_out=False, add_y_out=False, num_semantic_classes=1, ctrl_add_inp=True,
      ctrl_add_canvas=True, ctrl_add_d_out=False, ctrl_add_y_out=False, attn_add_inp=True,
      attn_add_canvas=True, attn_add_d_out=False, attn_add_y_out=False,
      ctrl_cnn_filter_size=[3] * 8, ctrl_cnn_depth=[8, 8, 16, 16, 32, 32, 64, 64],
      ctrl_cnn_pool=[1, 2, 1, 2, 1, 2, 2, 2], attn_cnn_filter_size=[3] * 6,
      attn_cnn_depth=[8, 8, 16, 16, 32, 32], attn_cnn_pool=[1, 2, 1, 2, 1, 2],
      attn_dcnn_filter_size=[3] * 7, attn_dcnn_depth=[32, 32, 16, 16, 8, 8, 1],
      attn_dcnn_pool=[2, 1, 2, 1, 2, 1, 1], attn_cnn_skip='1,1,1')
  if arch in ('kitti', 'cityscapes'):  # run_kitti.sh:68-111, run_cityscapes.sh:62-110
    opt.update(ctrl_cnn_depth=[16, 16, 32, 32, 64, 64, 64, 64], ctrl_cnn_pool=[2, 2, 1, 2, 1, 2, 1, 2],
               attn_cnn_depth=[16, 32, 32, 64, 64, 96], attn_dcnn_depth=[64, 64, 32, 32, 16, 16, 1],
               dynamic_var=True, fixed_gamma=False, add_skip_conn=True, add_d_out=True, add_y_out=True,
               attn_add_d_out=True, attn_add_y_out=True, ctrl_add_d_out=True, ctrl_add_y_out=True,
               attn_cnn_skip='1,0,1,0,1,0,1,0')
    if arch == 'cityscapes':
      opt.update(num_semantic_classes=9, fixed_gamma=True, use_iou_box=True)
  elif arch != 'cvppp':
    raise ValueError(arch)
  return opt


def encoder_flops_per_image(d):
  """sum over ctrl-CNN layers of 2*9*Cin*Cout*Hc*Wc."""
  h, w, tot, per = d['H'], d['W'], 0.0, []
  for i in range(d['ccnn_nlayers']):
    f = 2.0 * 9 * d['ccnn_channels'][i] * d['ccnn_channels'][i + 1] * h * w
    per.append(f)
    tot += f
    h, w = h // d['ccnn_pool'][i], w // d['ccnn_pool'][i]
  return tot, per


def seed_weights(model, seed):
  """Synthetic weights: the reference's truncated_normal(0.01) init is already in place
  (nnlib.py:53-54); give the BN EMA statistics random positive values and
  a plausible box so the Gaussian bands have realistic extent.  Values do not affect speed."""
  g = torch.Generator().manual_seed(seed)
  for k in model.weight_keys():
    t = model[k]
    if k.endswith('_ema_var'):
      t.copy_(torch.empty(t.shape).uniform_(0.5, 1.5, generator=g))
    elif k.endswith('_ema_mean'):
      t.copy_(torch.empty(t.shape).normal_(0.0, 0.05, generator=g))
  b = model['ctrl_mlp_b_0'].cpu()
  b[2:4] = float(np.log(0.35))
  model['ctrl_mlp_b_0'].copy_(b)


ROUNDS = ('r06', 'r05', 'r04', 'r03', 'r02', 'r01')


def pmc_traffic(images, size, name='rNN_pmc_encoder_traffic.json', want_source=False):
  """HBM bytes per launch group measured with rocprofv3 --pmc (separate FETCH_SIZE and WRITE_SIZE
  passes over `bench.py --pmc-group REPS [--pmc-which attn]`, summarised by tools/pmc_traffic.py);
  the newest committed round's file that matches this shape.  want_source: also which file, and whether the library
  and bench.py it was collected with are the ones running now (the counters are replayed, not measured in this run:
  gpurun refuses --pmc next to a trace, and a --pmc pass is its own process)."""
  import hashlib
  for rnd in ROUNDS:
    path = os.path.join(ROOT, 'profiles', rnd + name[3:])
    if not os.path.exists(path):
      continue
    rec = json.load(open(path))
    if rec.get('images') == images and rec.get('size') == size:
      if not want_source:
        return rec['hbm_bytes_per_launch_group']
      sha = lambda p: hashlib.sha256(open(p, 'rb').read()).hexdigest()[:16] if os.path.exists(p) else None
      now = {'bench_py_sha16': sha(os.path.join(ROOT, 'bench.py')),
             'librecattend_sha16': sha(os.path.join(ROOT, 'rec-attend-public_amd', 'librecattend.so'))}
      got = rec.get('collected_with')
      return rec['hbm_bytes_per_launch_group'], {
          'file': 'profiles/' + os.path.basename(path), 'collected_with': got,
          'same_library_as_this_run': None if not got else got.get('librecattend_sha16') == now['librecattend_sha16']}
  return (None, None) if want_source else None


def mfma_busy():
  """Group MFMA-busy fraction of the controller-CNN kernels from the newest committed SQ counter pass
  (profiles/r0x_pmc_sq_mfma_per_kernel.csv: per-kernel means per dispatch): sum of SQ_VALU_MFMA_BUSY_CYCLES over
  sum of GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs, each kernel weighted by its launches per timestep."""
  import csv
  for rnd in ROUNDS:
    path = os.path.join(ROOT, 'profiles', rnd + '_pmc_sq_mfma_per_kernel.csv')
    if not os.path.exists(path):
      continue
    rows = [r for r in csv.DictReader(open(path)) if any(k in r['kernel'] for k in ('ra::cpair::', 'ra::wino::', 'ra::csplit::', 'ra::conv::conv3x3_mfma<16, 1, 4, 2, 1, false'))
            and 'conv_pair8_mfma<4, false' not in r['kernel']]
    if not rows:
      continue
    base = min(int(r['dispatches']) for r in rows)
    busy = sum(float(r['SQ_VALU_MFMA_BUSY_CYCLES']) * round(int(r['dispatches']) / base) for r in rows)
    avail = sum(float(r['GRBM_GUI_ACTIVE']) / 8.0 * 1024.0 * round(int(r['dispatches']) / base) for r in rows)
    return {'frac': busy / avail, 'source': 'profiles/%s_pmc_sq_mfma_per_kernel.csv' % rnd,
            'per_kernel': {r['kernel'].split('(')[0].replace('void ', ''): float(r['SQ_VALU_MFMA_BUSY_CYCLES']) / (float(r['GRBM_GUI_ACTIVE']) / 8.0 * 1024.0)
                           for r in rows}}
  return None


PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_16x16x32_bf16: 16 384 FLOP in 16 cycles of a SIMD)
MFMA_CLOCK_GHZ = 2.4


def mfma_executed(enc_us=None):
  """What the matrix pipes EXECUTED in the controller-CNN launch group, from the newest committed SQ counter pass that holds
  SQ_INSTS_VALU_MFMA_MOPS_F32 / _BF16 (rocprofv3: add-or-multiply operations / 512, per dispatch): the time the group's MFMA
  instructions occupy their pipe at its dense peak — float32 ops at 157.3 TFLOP/s plus bf16 ops at 2 500 TFLOP/s (= MFMA
  instructions issued x cycles each / (1024 SIMDs x clock)) — over the group's duration.  Unlike roofline.frac (ALGORITHMIC FLOPs
  over the float32 peak, which the six-piece bf16 forms and Winograd can push past 1) this cannot exceed 1: it is the fraction
  of the launch group's time during which the matrix pipes had work."""
  import csv
  for rnd in ROUNDS:
    path = os.path.join(ROOT, 'profiles', rnd + '_pmc_sq_mfma_per_kernel.csv')
    if not os.path.exists(path):
      continue
    rows = [r for r in csv.DictReader(open(path)) if any(k in r['kernel'] for k in ('ra::cpair::', 'ra::wino::', 'ra::csplit::', 'ra::conv::conv3x3_mfma<16, 1, 4, 2, 1, false'))
            and 'conv_pair8_mfma<4, false' not in r['kernel']]
    if not rows or 'SQ_INSTS_VALU_MFMA_MOPS_BF16' not in rows[0]:
      continue
    base = min(int(r['dispatches']) for r in rows)
    per, tot_s, tot_cyc = {}, 0.0, 0.0
    for r in rows:
      w = round(int(r['dispatches']) / base)
      f32, b16 = 512.0 * float(r['SQ_INSTS_VALU_MFMA_MOPS_F32']), 512.0 * float(r['SQ_INSTS_VALU_MFMA_MOPS_BF16'])
      pipe_s = f32 / (PEAK_F32_MFMA_TFLOPS * 1e12) + b16 / (PEAK_BF16_MFMA_TFLOPS * 1e12)
      cyc = float(r['GRBM_GUI_ACTIVE']) / 8.0
      per[r['kernel'].split('(')[0].replace('void ', '')] = {
          'launches_per_timestep': w, 'gflop_f32_pipe': f32 / 1e9, 'gflop_bf16_pipe': b16 / 1e9, 'pipe_us_at_peak': pipe_s * 1e6,
          'frac_of_its_own_duration': pipe_s * MFMA_CLOCK_GHZ * 1e9 / cyc}
      tot_s += w * pipe_s
      tot_cyc += w * cyc
    out = {'pipe_us_at_peak_per_launch_group': tot_s * 1e6, 'frac_counter_clock': tot_s * MFMA_CLOCK_GHZ * 1e9 / tot_cyc,
           'peaks': {'f32_mfma_tflops': PEAK_F32_MFMA_TFLOPS, 'bf16_mfma_tflops': PEAK_BF16_MFMA_TFLOPS},
           'source': 'profiles/%s_pmc_sq_mfma_per_kernel.csv' % rnd, 'per_kernel': per,
           'note': 'executed matrix work: SQ_INSTS_VALU_MFMA_MOPS_{F32,BF16} x 512 FLOP each at the dense peak of its own pipe, over '
                   'the launch time (frac: this run\'s HIP-event time of the group; frac_counter_clock: the counter pass\'s own '
                   'GRBM_GUI_ACTIVE at %.1f GHz); bounded by 1, unlike roofline.frac' % MFMA_CLOCK_GHZ}
    if enc_us:
      out['frac'] = tot_s * 1e6 / enc_us
    return out
  return None


def _cgroup_cpus():
  """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a
  container often sees every host core in os.cpu_count() but is scheduled on far fewer)."""
  n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
  try:
    quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
    if quota != 'max':
      n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
  except Exception:
    pass
  return max(1, n)


def _cpu_feeds(opt, rng, n):
  """x [, d_in, y_in] for the CPU restatement (SURVEY 8d's synthetic inputs)."""
  H, W = opt['inp_height'], opt['inp_width']
  x = rng.rand(n, H, W, 3).astype(np.float32)
  if not opt.get('add_d_out'):
    return x, None, None
  d_in = np.eye(8, dtype=np.float32)[rng.randint(0, 8, (n, H, W))]
  lg = rng.randn(n, H, W, opt['num_semantic_classes']).astype(np.float32)
  e = np.exp(lg - lg.max(-1, keepdims=True))
  return x, d_in, (e / e.sum(-1, keepdims=True)).astype(np.float32)


def cpu_baseline(opt, seed, budget_s=12.0, batch=2, steps=4, arch='CVPPP'):
  """CPU stand-in baseline as BASELINE.md §3 / SURVEY.md §8(d) define it: the PyTorch-CPU
  restatement of the same graph (oracle/ra_oracle_torch.py: F.conv2d / conv_transpose2d /
  max_pool2d / matmul, i.e. oneDNN + BLAS), float32, on a bounded sample of the same workload —
  forwards of `batch` images x `steps` timesteps at the full cfg2 resolution (a timestep costs the
  same wherever it sits in the sequence).  Thread count: every usable core is tried first on ONE
  image-timestep; oversubscribed intra-op pools can be orders of magnitude slower than a moderate
  one on this graph's small tensors, so 32 threads are tried too and the faster setting is the one
  timed and reported (`cores` = threads actually used).  The reference's own TF-0.12 CPU path
  cannot run here, hence kind = "port"."""
  sys.path.insert(0, os.path.join(ROOT, 'oracle'))
  import ra_oracle as ora
  import ra_oracle_torch as ort
  usable = _cgroup_cpus()
  ort.set_dtype(torch.float32)
  H, W = opt['inp_height'], opt['inp_width']
  rng = np.random.RandomState(seed)
  try:
    with torch.no_grad():
      o1 = dict(opt)
      o1['timespan'] = 1
      P1 = ora.random_params(o1, seed)
      x1, d1, y1 = _cpu_feeds(opt, rng, 1)
      probe = {}
      for nt in sorted({usable, min(usable, 32)}, reverse=True):
        torch.set_num_threads(nt)
        ort.forward(o1, P1, x1, d1, y1)  # warm-up: thread pool, oneDNN primitive cache
        t0 = time.perf_counter()
        ort.forward(o1, P1, x1, d1, y1)
        probe[nt] = time.perf_counter() - t0
      threads = min(probe, key=probe.get)
      torch.set_num_threads(threads)
      o2 = dict(opt)
      o2['timespan'] = steps
      P2 = ora.random_params(o2, seed)
      x2, d2, y2 = _cpu_feeds(opt, rng, batch)
      n, t0 = 0, time.perf_counter()
      while True:
        ort.forward(o2, P2, x2, d2, y2)
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 128:  # bounded sample
          break
  finally:
    ort.set_dtype(torch.float64)
  return {'value': n * batch * steps / el, 'unit': 'instance-timesteps/s', 'cores': int(threads),
          'kind': 'port', 'label': 'CPU stand-in (PyTorch-CPU restatement, float32), %d threads of %d usable '
                                   'cores' % (threads, usable),
          'sample': 'PyTorch-CPU (oneDNN/BLAS) float32 restatement, %d forwards of B=%d x T=%d of the same '
                    '%dx%d %s-arch full_model graph in %.1f s on %d threads (one image-timestep probe: %s); stand-in '
                    'for the TF-0.12 CPU path, which cannot run here'
                    % (n, batch, steps, H, W, arch, el, threads,
                       ', '.join('%d threads %.2f s' % kv for kv in sorted(probe.items())))}


def graph_time_us_plain(fn, reps=20, inner=8):
  """Average duration of one `fn` launch group: `inner` copies captured in one HIP graph (a replay's fixed ~10 us must not
  be charged to the kernels), replayed `reps` times between two HIP events on the launch stream."""
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


def other_rooflines(model, feed, T, H, W, slot_eng=None):
  """`roofline` (controller-CNN launch group, MFMA), `roofline_attn` (extract + paste, HBM), tail and controller times of a
  full_model stage at cfg3 / cfg5, in the protocol of the cfg2 line: each group exactly as the engine issues it for ONE batch,
  captured alone in a HIP graph and timed with HIP events on the launch stream; `as_launched` = the same groups over the images
  a pipeline slot decodes per forward.  FLOPs / bytes: SURVEY 8(d) (2*9*Cin*Cout per conv output pixel; H*W*(C0+3)*4 B per
  image-timestep for the resample)."""
  import ra_ops as ops
  eng, d = model.engine, model.dims
  eng.forward(feed['x'], d_in=feed.get('d_in'), y_in=feed.get('y_in'))
  torch.cuda.synchronize()
  tot_f, per_f = encoder_flops_per_image(d)
  Fh, Fw = d['Fh'], d['Fw']
  pflags = (0 if d['disable_overwrite'] else ops.PASTE_Y_PREFILLED) | ops.PASTE_CANVAS_FLOORED

  def groups(e):
    sb, Wt = e.subs[0], e.W
    n = int(sb['img'].shape[0])

    def enc_step(step):
      src = sb['img'] if step[1] == 0 else sb['ccnn'][step[1] - 1]
      e._run_cnn([step], Wt['ccnn'], src, sb['ccnn'], 1, 'ctrl_cnn', plane=sb['canvas'], cache=sb.get('l0cache'))

    def attn_group():
      ops.extract_direct(sb['img'], 0, sb['attn'][0], Fh, Fw, d['C0p'], True, sb['x_patch'][0], canvas=sb['canvas'], canvas_chan=d['D'])
      ops.paste_direct(sb['y_out_patch'][0], 0, sb['attn'][0], -5.0, d['disable_overwrite'], sb['y_out'].data_ptr(), T * H * W, H, W,
                       canvas=sb['canvas'], flags=pflags)
    return sb, n, enc_step, attn_group

  sb, n, enc_step, attn_group = groups(eng)
  layers = []
  for step in eng.plan['ccnn']:
    us = graph_time_us_plain(lambda: enc_step(step))
    fl = sum(per_f[i] for i in step[1:])
    kind = ('pair' if step[0] == 'pair' else 'K1s' if eng.W['ccnn_split'][step[1]] is not None else
            'K1w' if eng.W['ccnn_wino'][step[1]] is not None and step[1] > 0 else 'K1')
    layers.append({'layers': list(step[1:]), 'kernel': kind, 'avg_us': us, 'gflop': fl * n / 1e9, 'tflops': fl * n / (us * 1e-6) / 1e12})
  enc_us = graph_time_us_plain(lambda: [enc_step(st_) for st_ in eng.plan['ccnn']])
  enc_bytes = 4.0 * sb['canvas'].numel()
  for st_ in eng.plan['ccnn']:
    src_ = sb['img'] if st_[1] == 0 else sb['ccnn'][st_[1] - 1]
    enc_bytes += 4.0 * (src_.numel() + sb['ccnn'][st_[-1]].numel())
  ach = tot_f * n / (enc_us * 1e-6) / 1e12
  roof = {'kernel': 'controller CNN: %d layers in %d launches per timestep per batch of %d images (%s)'
                    % (d['ccnn_nlayers'], len(eng.plan['ccnn']), n, ', '.join('L%s %s' % ('+'.join(map(str, l['layers'])), l['kernel']) for l in layers)),
          'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS,
          'traffic': None, 'algorithmic_bytes_per_launch_group': enc_bytes, 'flop_per_launch_group': tot_f * n,
          'flop_per_image_timestep': tot_f, 'avg_us_per_launch_group': enc_us, 'layers': layers,
          'peak_note': 'algorithmic FLOPs (SURVEY 8d) over the dense f32-input MFMA peak; the K1s / K1w / split-pair launches do their '
                       'multiplies on the bf16 pipe as six piece products or as Winograd, so this is a speed relative to a '
                       'float32-MFMA-bound kernel, not a bounded fraction'}
  attn_us = graph_time_us_plain(attn_group)
  abytes = float(H * W * (d['acnn_channels'][0] + 3) * 4)
  roof_a = {'kernel': 'ra::attnd::extract_rows_kernel + paste_win_kernel, one batch of %d images' % n, 'bound': 'hbm',
            'achieved': abytes * n / (attn_us * 1e-6) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
            'frac': abytes * n / (attn_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 'traffic': None, 'bytes_per_launch_group': abytes * n,
            'bytes_per_image_timestep': abytes, 'avg_us_per_launch_group': attn_us,
            'note': 'algorithmic bytes (SURVEY 8d: H*W*(C0+3)*4 per image-timestep) over the two dependent launches; the kernels are '
                    'window-only, so the bytes really moved are a fraction of that (cfg2: 0.26x, profiles/r05_pmc_attn_traffic.json)'}
  extra = {'tail_us': graph_time_us_plain(lambda: eng._launch_tail(sb, 1, False, sb['ccnn'][-1]))}
  if slot_eng is not None and slot_eng.subs:
    sb2, n2, enc2, attn2 = groups(slot_eng)
    us2 = graph_time_us_plain(lambda: [enc2(st_) for st_ in slot_eng.plan['ccnn']])
    roof['as_launched'] = {'images_per_launch': n2, 'avg_us_per_launch_group': us2, 'achieved': tot_f * n2 / (us2 * 1e-6) / 1e12,
                           'frac': tot_f * n2 / (us2 * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS}
    sb2['attn'][0].copy_(sb['attn'][0][:1].expand(n2, -1))
    ua2 = graph_time_us_plain(attn2)
    roof_a['as_launched'] = {'images_per_launch': n2, 'extract_paste_us': ua2, 'frac_algorithmic': abytes * n2 / (ua2 * 1e-6) / 1e9 / PEAK_HBM_GBS}
    extra['tail_us_as_launched'] = graph_time_us_plain(lambda: slot_eng._launch_tail(sb2, 1, False, sb2['ccnn'][-1]))
  return roof, roof_a, extra


OTHER_CONFIGS = {  # BASELINE.json configs[2] / configs[4] at the sizes the reference feeds the model
    'cfg3': dict(arch='kitti', H=128, W=448, T=20, B=16, stages=('box_model', 'full_model'),
                 what='KITTI arch 128x448 (1242x375 crops resized as the reference does), box_model + full_model two-stage, T=20, B=16'),
    'cfg5': dict(arch='cityscapes', H=256, W=512, T=20, B=4, stages=('full_model',),
                 what='Cityscapes arch 256x512 (2048x1024 tiles resized), full_model, T=20, 4 images per GPU of a batch of 32 over 8 GPUs'),
}


def bench_other(args, rank, world, name):
  """The other single-GPU-sized BASELINE.json configurations as driver-reproducible lines
  (`--config cfg3|cfg5`): eval forward of every stage on synthetic inputs, same protocol as cfg2."""
  import box_model
  import full_model
  import ra_dist
  c = OTHER_CONFIGS[name]
  B, T, H, W = c['B'], c['T'], c['H'], c['W']
  opt = make_opt(c['arch'], H, W, T)
  g = torch.Generator().manual_seed(1234 + rank)
  feed = {'x': torch.rand((B, H, W, 3), generator=g).cuda(), 'phase_train': False}
  nc = opt['num_semantic_classes']
  feed['d_in'] = torch.nn.functional.one_hot(torch.randint(0, 8, (B, H, W), generator=g), 8).float().cuda()
  feed['y_in'] = torch.softmax(torch.randn((B, H, W, nc), generator=g), dim=-1).cuda()
  yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
  y_gt = torch.zeros((B, T, H, W))
  for t in range(min(T, 6)):
    y_gt[:, t] = (((yy - (20 + 15 * t) % H) ** 2 / 400.0 + (xx - (30 + 60 * t) % W) ** 2 / 900.0) <= 1).float()
  models = []
  for st in c['stages']:
    m = (box_model.get_model(opt) if st == 'box_model' else full_model.get_model(opt, is_training=False))
    seed_weights(m, 1234 + rank)
    models.append((st, m))

  # every stage decodes through its own DecodePipeline (full_model.py): `--in-flight` batches (or parts
  # of --part-images images) decode concurrently over all stages, each on its own HIP graph + stream
  feed['y_gt'] = y_gt.cuda()
  per = args.part_images or B
  parts = -(-B // per)
  co = max(1, args.coalesce)  # submitted batches one slot decodes as ONE forward (DecodePipeline(coalesce=...))
  depth = max(parts, max(1, args.in_flight) // len(models) // co)
  per_stage = max(1, args.streams // len(models)) if args.streams else None  # HIP streams per stage (default min(depth, 4))
  pipes = [(st, m.pipeline(depth, max_images=per * co, co_resident=min(depth, per_stage or 4) * len(models), streams=per_stage, coalesce=co))
           for st, m in models]

  def step():
    for st, pipe in pipes:
      while pipe.full(B):
        pipe.retire()
      pipe.submit(['attn_box', 's_out'] if st == 'box_model' else ['y_out', 's_out'], feed)

  def drain():
    for _, pipe in pipes:
      pipe.drain()

  for _ in range(max(args.warmup, 2 * depth * co)):
    step()
  drain()
  ra_dist.barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
  drain()
  ra_dist.barrier()
  elapsed = ra_dist.max_over_ranks(time.perf_counter() - t0)
  for _, pipe in pipes:
    for eng_k, _ in pipe.slots:
      eng_k.check_status(recover=False)
  if rank == 0:
    fm = [m for st, m in models if st == 'full_model'][0]
    fpipe = [p for st, p in pipes if st == 'full_model'][0]
    roof, roof_a, extra = other_rooflines(fm, feed, T, H, W, slot_eng=fpipe.slots[0][0] if fpipe.slots else None)
    line = {
        'metric': 'instance-timesteps/sec, %s (whole job)' % name, 'value': world * B * T * args.steps / elapsed,
        'unit': 'instance-timesteps/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s: %s' % (name, c['what']), 'arch': c['arch'], 'H': H, 'W': W, 'T': T,
                   'batch_per_gpu': B, 'stages': list(c['stages']), 'parts_per_batch': parts,
                   'parts_in_flight': depth * len(pipes) * co, 'batches_per_launch': co,
                   'controller': ('group-shared (16 workgroups per %d images)' % ops_group(pipes[-1][1].slots[0][0]) if pipes[-1][1].slots[0][0].subs[0].get('ctrl_batch')
                                  else 'split (16 workgroups per image)' if 'ctrl_ws' in pipes[-1][1].slots[0][0].subs[0]
                                  else 'single workgroup per image')},
        'roofline': roof, 'roofline_attn': roof_a}
    line.update(extra)
    if world == 1 and not args.no_cpu_baseline:
      line['cpu_baseline'] = cpu_baseline(opt, 1234, arch=c['arch'])
      line['cpu_baseline']['note'] = 'full_model stage only (the restatement of box_model\'s teacher-forced loop is not timed)' if len(models) > 1 else 'full_model'
    print(json.dumps(line))
  if world > 1:
    ra_dist.barrier()
    torch.distributed.destroy_process_group()


def ops_group(eng):
  import ra_ops
  return ra_ops.ctrl_batch_group(eng.desc, eng.subs[0]['img'].shape[0])


def train_steps(rank, world, B, T, S, steps, warmup, sync_bn=False, dtype='f32'):
  """`steps` timed optimisation steps at BASELINE.json configs[3]'s per-GPU shapes (every rank takes part: the
  gradient all-reduce is inside the step).  Returns (elapsed seconds (max over ranks), last loss, model)."""
  import full_model
  import full_model_train as fmt
  import ra_dist
  opt = make_opt('cvppp', S, S, T)
  opt.update(use_knob=True, knob_base=1.0, knob_decay=0.9, steps_per_knob_decay=300, knob_box_offset=300,
             knob_segm_offset=500, knob_use_timescale=True, gt_box_ctr_noise=0.05, gt_box_pad_noise=0.1,
             gt_segm_noise=0.3, base_learn_rate=1e-3, learn_rate_decay=0.96, steps_per_learn_rate_decay=5000,
             sync_bn=bool(sync_bn), seed=1234, compute_dtype='bf16' if dtype == 'bf16' else 'float32')
  model = full_model.get_model(opt, is_training=True)
  rng = np.random.RandomState(1234 + rank)
  x, y_gt, s_gt = fmt.synthetic_batch(rng, B, S, S, T)
  gen = torch.Generator(device='cuda').manual_seed(1234 + rank)
  feed = {'x': torch.as_tensor(x).cuda(), 'y_gt': torch.as_tensor(y_gt).cuda(), 's_gt': torch.as_tensor(s_gt).cuda(),
          'phase_train': True, 'generator': gen}
  for _ in range(max(warmup, 2)):  # the first step of a shape runs eagerly, the second is captured
    loss, _ = model.run(['loss', 'train_step'], feed)
  ra_dist.barrier()
  t0 = time.perf_counter()
  for _ in range(steps):
    loss, _ = model.run(['loss', 'train_step'], feed)
  ra_dist.barrier()
  dt = ra_dist.max_over_ranks(time.perf_counter() - t0)
  model.trainer.flush_status()  # run() checks a step's solver statuses one step late: the last step's now
  return dt, float(loss), model


def train_layer_bytes(opt, B, S, T, bf16_store=False):
  """HBM bytes the conv layers' passes of ONE training step move when every pass reads and writes each of its tensors
  exactly once (float32 tensors in both dtypes — the bf16 mode rounds operands on chip): per layer call
  forward conv (X in, U out) + normalise/ReLU/pool (U in, Y out); backward BatchNorm sums (U, dY in) + its input
  gradient (U, dY in, dU out) + data gradient (dU in, dX out; not for a layer fed by data) + filter gradient (X, dU in)
  = 3 X + 7 U + 3 Y (2 X + 6 U + 3 Y without the data gradient).  The three networks of a timestep (full_model.py:455-535)
  times T timesteps; the controller, the attention resamples and the loss head are not counted (launch-bound, small).
  bf16_store (the bf16 mode's stacked step): U, and Y of every layer but a net's last, are 2 bytes per element where the
  float4 BatchNorm kernels take the channel count (a multiple of 4, a quarter of it a power of two); a net's first input
  (the packed image, the extracted patch, the attention CNN's last output) stays float32."""
  r4 = lambda c: -(-c // 4) * 4
  total = 0
  st_ok = lambda c: bf16_store and c % 4 == 0 and ((c // 4) & (c // 4 - 1)) == 0 and c // 4 <= 64

  def net(h, w, cin, depths, pools, transposed, first_has_dgrad):
    nonlocal total
    sx = 4  # bytes per element of the layer's input
    for i, (co, p) in enumerate(zip(depths, pools)):
      if transposed:
        h, w = h * p, w * p
      su = 2 if st_ok(co) else 4
      sy = 2 if (st_ok(co) and i != len(depths) - 1) else 4
      X = B * (h // (p if transposed else 1)) * (w // (p if transposed else 1)) * r4(cin) * sx
      U = B * h * w * co
      if not transposed:
        Y = U // (p * p)
        h, w = h // p, w // p
      else:
        Y = U
      U, Y = U * su, Y * sy
      total += (3 * X + 7 * U + 3 * Y) if (i or first_has_dgrad) else (2 * X + 6 * U + 3 * Y)
      cin, sx = co, sy
    return h, w, cin
  net(S, S, 4, opt['ctrl_cnn_depth'], opt['ctrl_cnn_pool'], False, False)
  F = opt['filter_height']
  h, w, c = net(F, opt['filter_width'], 4, opt['attn_cnn_depth'], opt['attn_cnn_pool'], False, True)
  net(h, w, c, opt['attn_dcnn_depth'], opt['attn_dcnn_pool'], True, True)
  return total * T


def train_roofline(opt, B, S, T, ms_per_step, bf16_store=False):
  nbytes = train_layer_bytes(opt, B, S, T, bf16_store)
  ach = nbytes / (ms_per_step * 1e-3) / 1e9
  return {'bound': 'hbm', 'achieved': ach, 'peak': 8000.0, 'unit': 'GB/s', 'frac': ach / 8000.0,
          'bytes_per_step': nbytes,
          'bf16_storage': bool(bf16_store),
          'note': 'conv-layer passes only, each tensor of a pass moved once (3 X + 7 U + 3 Y per layer call, DESIGN.md §4 '
                  'training kernels; with bf16 storage U and Y are 2 bytes per element); the whole step time in the denominator: '
                  'the step is launch-latency bound on the patch-sized layers, not HBM bound'}


DTYPE_NOTE = {'f32': 'float32 throughout',
              'bf16': 'conv forward / data / filter gradients with bf16 operands on the bf16 MFMA, float32 accumulation, and '
                      'the conv layers\' U / Y / dY / dU stored as bf16 between their passes; BatchNorm statistics, master weights, '
                      'Adam state, the controller, the attention resamples and the loss float32'}


def train_object(rank, world, B, T, S, steps=3, dtype='f32'):
  """The `train` object of the default line: the training step timed by the same process."""
  import ra_dist
  elapsed, loss, model = train_steps(rank, world, B, T, S, steps, 2, dtype=dtype)
  obj = {'workload': 'cfg4 shapes: CVPPP-arch full_model TRAINING step, %dx%d, T=%d, B=%d per GPU (global %d), use_knob, in-graph '
                     'random crop, data-parallel with one flat-bucket all-reduce; %s' % (S, S, T, B, B * world, DTYPE_NOTE[dtype]),
         'steps': steps, 'ms_per_step': 1e3 * elapsed / steps, 'value': world * B * T * steps / elapsed,
         'unit': 'instance-timesteps/s', 'dtype': dtype, 'final_loss': loss, 'grad_bucket_floats': int(model.trainer.bucket.n),
         'hip_graph': bool(model.trainer.use_graph), 'fused_controller': getattr(model.trainer, '_ctl', None) is not None,
         'ranks_in_communicator': ra_dist.comm_size(),
         'roofline': train_roofline(model.opt, B, S, T, 1e3 * elapsed / steps, bool(getattr(model.trainer, 'bf16_store', False)))}
  del model
  torch.cuda.empty_cache()
  return obj


def bench_train(args, rank, world, B, T, S):
  """One step = augmentation + forward (BN batch statistics, GT knobs) + both matchings + backward + gradient
  all-reduce + clip/Adam on B synthetic CVPPP-shaped images per GPU.  --dtype f32 (the reference's arithmetic) or
  bf16 (BASELINE.json configs[3]: model_opt['compute_dtype'] = 'bf16')."""
  import ra_dist
  elapsed, loss, model = train_steps(rank, world, B, T, S, args.steps, args.warmup, sync_bn=args.sync_bn, dtype=args.dtype)
  if rank == 0:
    print(json.dumps({
        'metric': 'training instance-timesteps/sec (forward + backward + all-reduce + Adam), whole job',
        'value': world * B * T * args.steps / elapsed, 'unit': 'instance-timesteps/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
        'config': {'workload': 'cfg4 shapes: CVPPP-arch full_model TRAINING step, %dx%d, T=%d, B=%d per GPU '
                               '(global %d), use_knob, data-parallel with one flat-bucket all-reduce; %s'
                               % (S, S, T, B, B * world, DTYPE_NOTE[args.dtype]),
                   'batch_per_gpu': B, 'global_batch': B * world, 'parallelism': 'dp%d' % world,
                   'grad_bucket_floats': int(model.trainer.bucket.n),
                   'bn_moments': 'whole batch (sync_bn)' if model.trainer.sync_bn else 'per-rank shard',
                   'ranks_in_communicator': ra_dist.comm_size()},
        'roofline': train_roofline(model.opt, B, S, T, 1e3 * elapsed / args.steps, bool(getattr(model.trainer, 'bf16_store', False))),
        'final_loss': float(loss)}))
  if world > 1:
    ra_dist.barrier()
    torch.distributed.destroy_process_group()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=200)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--batch', type=int, default=8, help='images per GPU')
  ap.add_argument('--size', type=int, default=512)
  ap.add_argument('--timespan', type=int, default=16)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-graph', action='store_true')
  ap.add_argument('--part-images', type=int, default=0, help='--config cfg3|cfg5: cut each batch into parts of this many images')
  ap.add_argument('--wino-unfuse', action='store_true', help='tuning aid: L2 direct + L3 Winograd instead of the fused L2+L3 pair')
  ap.add_argument('--no-pair-wino', action='store_true', help='tuning aid: direct second layer in the fused L2+L3 pair')
  ap.add_argument('--no-wino', action='store_true', help='tuning aid: direct conv for every controller-CNN layer')
  ap.add_argument('--fuse-patch-pairs', action='store_true', help='tuning aid: fused two-layer launches in the patch-sized nets too')
  ap.add_argument('--no-ctrl-split', action='store_true', help='tuning aid: one-workgroup-per-image controller')
  ap.add_argument('--streams', type=int, default=0, help='HIP streams the in-flight batches share (default: min(in-flight, 4))')
  ap.add_argument('--in-flight', type=int, default=0,
                  help='batches decoding concurrently per GPU (DecodePipeline depth; 1 = one after the other; '
                       'default: 8 (at cfg2 on 4 streams; at cfg3 over its two stages))')
  ap.add_argument('--coalesce', type=int, default=0,
                  help='consecutively submitted batches one pipeline slot decodes as ONE forward (DecodePipeline(coalesce=...)); '
                       'default: 2 at cfg2 (four slots of 2 x 8 images for the 8 batches in flight), 2 at cfg3 (2 x 16), 4 at cfg5 (4 x 4)')
  ap.add_argument('--nsub', type=int, default=0, help='stream-parallel sub-batches (0 = auto)')
  ap.add_argument('--no-fuse-score', action='store_true', help='tuning aid: score MLP as its own launch')
  ap.add_argument('--host-output', action='store_true',
                  help='also copy y_out + s_out of every batch to pinned host memory (PCIe inclusive; not the headline value)')
  ap.add_argument('--host-input', action='store_true',
                  help='hand x over as a pinned HOST buffer each step (PCIe-inclusive rate; never the headline value)')
  ap.add_argument('--pmc-group', type=int, default=0, metavar='REPS',
                  help='profiling aid: after one forward, launch only the encoder group REPS times '
                       'eagerly and exit (run under rocprofv3 --pmc; see tools/pmc_traffic.py)')
  ap.add_argument('--pmc-which', default='enc', choices=['enc', 'attn', 'tail'],
                  help='which launch group --pmc-group repeats: the encoder, extract+paste, or the whole tail')
  ap.add_argument('--no-cache-first', action='store_true', help='tuning aid: recompute the whole first layer per timestep')
  ap.add_argument('--no-prefill-ride', action='store_true', help='tuning aid: the per-forward y_out prefill as its own launch instead of riding on the first controller-CNN launch')
  ap.add_argument('--config', default='cfg2', choices=['cfg2', 'cfg3', 'cfg5'],
                  help='cfg2 (default) = the headline workload; cfg3 / cfg5 = the KITTI two-stage and Cityscapes '
                       'configurations of BASELINE.json as their own JSON lines')
  ap.add_argument('--sync_bn', action='store_true', help='--train: BatchNorm moments over the whole data-parallel batch')
  ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'], help="--train: the conv layers' operand type (model_opt['compute_dtype'])")
  ap.add_argument('--no-train-object', action='store_true', help='skip the `train` object (3 timed training steps) of the default line')
  ap.add_argument('--train', action='store_true',
                  help='time the TRAINING step instead (BASELINE.json configs[3] shapes: B images per GPU, '
                       'data-parallel, one RCCL all-reduce of the gradient bucket per step); prints its own JSON line')
  args = ap.parse_args()
  if args.in_flight <= 0:
    args.in_flight = {'cfg2': 8, 'cfg3': 8, 'cfg5': 8}[args.config]  # eight submitted batches decoding, as at cfg2
  if args.coalesce <= 0 and args.config != 'cfg2':
    # cfg3 / cfg5 (round 5, second session): coalesced slots as at cfg2 — 2 x 16 images per forward at cfg3, 4 x 4 at cfg5
    # (32.3k -> 38.2k and 17.9k -> 30.7k instance-timesteps/s; `--coalesce 1 --in-flight 6|2` = the protocol of rounds 2-5a)
    args.coalesce = {'cfg3': 2, 'cfg5': 4}[args.config]

  import ra_dist
  if int(os.environ.get('WORLD_SIZE', '1')) == 1 and args.gpus > 1:
    raise SystemExit('launch with torch.distributed.run --nproc-per-node %d' % args.gpus)
  torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
  rank, world, local_rank = ra_dist.init('nccl')  # 'nccl' is RCCL on ROCm

  if args.config != 'cfg2':
    return bench_other(args, rank, world, args.config)
  import full_model
  B, T, S = args.batch, args.timespan, args.size
  opt = make_opt('cvppp', S, S, T)
  model = full_model.get_model(opt, is_training=False)
  seed_weights(model, 1234 + rank)
  eng = model.engine
  eng.use_graph = not args.no_graph
  eng.nsub = args.nsub
  eng.fuse_score = not args.no_fuse_score
  eng.ctrl_split = not args.no_ctrl_split
  eng.use_wino = not args.no_wino
  eng.fuse_patch_pairs = args.fuse_patch_pairs
  eng.wino_unfuse = args.wino_unfuse
  eng.pair_wino = not args.no_pair_wino
  eng.cache_first = not args.no_cache_first
  eng.prefill_ride = not args.no_prefill_ride
  g = torch.Generator().manual_seed(1234 + rank)
  x = torch.rand((B, S, S, 3), generator=g, dtype=torch.float32).cuda()
  if args.host_input:
    x = x.cpu().pin_memory()
  feed = {'x': x, 'phase_train': False}

  barrier = ra_dist.barrier

  if args.train:
    return bench_train(args, rank, world, B, T, S)

  if args.pmc_group:
    eng.forward(feed['x'])
    sb = eng.subs[0]
    import ra_ops as ops
    d = model.dims
    for _ in range(args.pmc_group):
      if args.pmc_which == 'enc':
        eng._run_cnn(eng.plan['ccnn'], eng.W['ccnn'], sb['img'], sb['ccnn'], 1, 'ctrl_cnn',
                     plane=sb['canvas'], cache=sb.get('l0cache'))
      elif args.pmc_which == 'attn':
        ops.extract_direct(sb['img'], 0, sb['attn'][0], d['Fh'], d['Fw'], d['C0p'], True, sb['x_patch'][0],
                           canvas=sb['canvas'], canvas_chan=d['D'])
        ops.paste_direct(sb['y_out_patch'][0], 0, sb['attn'][0], -5.0, d['disable_overwrite'],
                         sb['y_out'].data_ptr(), T * S * S, S, S, canvas=sb['canvas'],
                         flags=ops.PASTE_Y_PREFILLED | ops.PASTE_CANVAS_FLOORED)
      else:
        eng._launch_tail(sb, 1, False, sb['ccnn'][-1])
    torch.cuda.synchronize()
    print(json.dumps({'pmc_group': args.pmc_group, 'images': int(sb['img'].shape[0]), 'size': S}))
    return

  # one batch alone, launch to completion (the latency a lone model.run sees; SURVEY.md §8d's literal protocol: median
  # of >= 20 forwards, each bracketed by a device synchronize, after >= 5 warm-ups)
  for _ in range(max(args.warmup, 5)):
    eng.forward(feed['x'])
  torch.cuda.synchronize()
  lone = []
  for _ in range(21):
    t0 = time.perf_counter()
    eng.forward(feed['x'])
    torch.cuda.synchronize()
    lone.append(1e3 * (time.perf_counter() - t0))
  lone_ms = float(np.median(lone))

  # the timed region: K steps = K batches through the evaluator's decode pipeline
  # (full_model.DecodePipeline: up to --in-flight batches decode concurrently, each a whole
  # forward of B images on its own HIP graph + stream; every step is a complete forward)
  coalesce = args.coalesce if args.coalesce > 0 else (2 if (args.in_flight >= 2 and B <= 8) else 1)
  pipe = model.pipeline(max(1, args.in_flight // coalesce), streams=args.streams or None, coalesce=coalesce)

  def step():
    while pipe.full():
      pipe.retire()
    pipe.submit(['y_out', 's_out'], feed, to_host=args.host_output)

  for _ in range(max(args.warmup, 2 * pipe.depth * coalesce)):  # every slot allocates + captures its graph
    step()
  pipe.drain()
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
  pipe.drain()
  barrier()
  elapsed = ra_dist.max_over_ranks(time.perf_counter() - t0)
  value = world * B * T * args.steps / elapsed
  # SURVEY 8(d)'s protocol beside the contract's mean: the MEDIAN time per batch in the steady state, fill and drain of the
  # pipeline excluded — the host clock at every completion (retire() returns when the oldest batch has finished); a
  # window of `streams` consecutive completions is one batch per stream, so (t[i + w] - t[i]) / w is a per-batch time
  n_steady = max(24, 3 * pipe.depth * coalesce)
  done = []
  for _ in range(pipe.depth * coalesce):
    pipe.submit(['y_out', 's_out'], feed, to_host=args.host_output)
  for _ in range(n_steady):
    while pipe.full():
      pipe.retire()
      done.append(time.perf_counter())
    pipe.submit(['y_out', 's_out'], feed, to_host=args.host_output)
  pipe.drain()
  wdw = max(1, int(pipe.streams)) * coalesce  # one launch per stream: that many batches complete per round of the streams
  gaps = [(done[i + wdw] - done[i]) / wdw for i in range(wdw, len(done) - wdw)]
  steady_ms = 1e3 * ra_dist.max_over_ranks(float(np.median(gaps)))
  for eng_k, _ in pipe.slots:  # a controller workgroup that timed out on its peers would have produced garbage: fail loudly
    eng_k.check_status(recover=False)

  out = {
      'metric': 'instance-timesteps/sec, full_model forward 512x512 T=16 (whole job; pipelined throughput: %d batches in flight '
                'per GPU, fill and drain of the pipeline inside the timed region; config.lone_batch_* = one sync-bracketed forward)' % (pipe.depth * coalesce),
      'value': value, 'unit': 'instance-timesteps/s', 'per_gpu': value / world,
      'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
      'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'config': {'workload': 'cfg2: CVPPP-arch full_model eval forward, %dx%d, T=%d, B=%d per GPU, '
                             'y_out+s_out' % (S, S, T, B),
                 'arch': 'cvppp', 'H': S, 'W': S, 'T': T, 'batch_per_gpu': B,
                 'global_batch': B * world, 'parallelism': 'batch-sharded x%d, no collective' % world,
                 'ranks_in_communicator': ra_dist.comm_size(),
                 'hip_graph': bool(eng.use_graph), 'tile_tickets': bool(eng.tile_tickets), 'batches_in_flight': pipe.depth * coalesce, 'pipeline_slots': pipe.depth,
                 'batches_per_launch': coalesce,
                 'pipeline_note': ('each of the %d pipeline slots decodes %d consecutively submitted batches of %d images as ONE forward '
                                   '(DecodePipeline(coalesce=%d): eval-mode images are independent, every batch is collected on its own); '
                                   '--coalesce 1 = one batch per slot, the protocol of rounds 2-4' % (pipe.depth, coalesce, B, coalesce)),
                 'steady_ms_per_step_median': steady_ms, 'steady_value_median': world * B * T / (steady_ms * 1e-3),
                 'steady_protocol': 'median over %d completions of the time per batch in the steady state of the pipeline '
                                    '(fill and drain excluded; windows of %d consecutive completions)' % (n_steady, wdw),
                 'lone_batch_ms': lone_ms, 'lone_batch_value': B * T / (lone_ms * 1e-3),
                 'lone_batch_protocol': 'median of 21 forwards, each bracketed by a device synchronize', 'input': 'host (PCIe inclusive)' if args.host_input else 'resident in HBM',
                 'output': 'y_out + s_out copied to pinned host memory (PCIe inclusive)' if args.host_output else 'left in HBM'},
  }

  train_obj = train_bf16 = None
  if not args.no_train_object:
    train_obj = train_object(rank, world, B, T, S)  # every rank: the step holds the gradient all-reduce
    train_bf16 = train_object(rank, world, B, T, S, dtype='bf16')
  if rank == 0:
    if train_obj is not None:
      out['train'] = train_obj
      out['train_bf16'] = train_bf16
    # ---- roofline objects: the launch groups exactly as the product issues them (one
    # sub-batch of Bs images), each captured alone in a HIP graph and replayed between two HIP
    # events on the launch stream, so host launch overhead does not pollute kernel time ----
    import ra_ops as ops
    d, Wt, sb = model.dims, eng.W, eng.subs[0]
    Bs = sb['img'].shape[0]
    Fh, Fw = d['Fh'], d['Fw']

    probe_tk = ops.tickets_alloc(256, sb['img'].device) if eng.tile_tickets else None
    probe_fill = {}

    def graph_time_us(fn, reps=20, inner=8, tickets=False):
      """Average duration of one `fn` launch group: `inner` copies are captured in one HIP graph
      (a replay has a fixed ~10 us cost that must not be charged to the kernels) and the graph is
      replayed `reps` times between two HIP events on the launch stream.  tickets: as in the engine's forward, the controller
      CNN's persistent launches draw their tiles (one zeroing fill of the ticket scratch per replay, inside the graph)."""
      fn()
      torch.cuda.synchronize()
      g = torch.cuda.CUDAGraph()
      with torch.cuda.graph(g):
        bound = tickets and probe_tk is not None and ops.tickets_bind(probe_tk)
        try:
          for _ in range(inner):
            fn()
        finally:
          if bound:
            ops.tickets_unbind()
      for _ in range(3):
        g.replay()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      torch.cuda.synchronize()
      e0.record()
      for _ in range(reps):
        g.replay()
      e1.record()
      torch.cuda.synchronize()
      us = 1e3 * e0.elapsed_time(e1) / (reps * inner)
      if tickets and probe_tk is not None:
        # the zeroing fill of the ticket scratch is one launch per FORWARD in the engine (16 timesteps x 6 launches) but one per
        # `inner` groups here: its share is taken out again
        if 'us' not in probe_fill:
          probe_fill['us'] = graph_time_us(lambda: ops.fill(probe_tk, 0.0))
        us -= probe_fill['us'] / inner
      return us

    rides = (not eng.box) and (not d['disable_overwrite']) and eng._prefill_rides(sb)

    def enc_step(step, tt_=1):  # tt_ = 1: the steady-state (cached) form; 0: the first timestep, which fills the cache
      first = step[1]           # ... and carries the once-per-forward prefill of y_out as a rider (ra_engine._launch_pack)
      src = sb['img'] if first == 0 else sb['ccnn'][first - 1]
      eng._run_cnn([step], Wt['ccnn'], src, sb['ccnn'], tt_, 'ctrl_cnn', plane=sb['canvas'],
                   cache=sb.get('l0cache'), fill=sb['y_out'] if (tt_ == 0 and first == 0 and rides) else None)

    tot_f, per_f = encoder_flops_per_image(d)
    layers = []
    for step in eng.plan['ccnn']:
      us = graph_time_us(lambda: enc_step(step), tickets=True)
      fl = sum(per_f[i] for i in step[1:])
      layers.append({'layers': list(step[1:]), 'fused': step[0] == 'pair', 'avg_us': us,
                     'gflop': fl * Bs / 1e9, 'tflops': fl * Bs / (us * 1e-6) / 1e12})
    enc_us = graph_time_us(lambda: [enc_step(st_) for st_ in eng.plan['ccnn']], tickets=True)
    cache_us = 0.0
    if 'l0cache' in sb:
      # the cache is filled by the FIRST timestep's launch (un-cached kernel + cache stores, valid while
      # the canvas is zero); what it costs over a cached launch is spread over the T timesteps
      ops.fill(sb['canvas'], 0.0)
      t0_us = graph_time_us(lambda: enc_step(eng.plan['ccnn'][0], 0), reps=10, inner=2)
      cache_us = max(0.0, t0_us - layers[0]['avg_us'])
      enc_us += cache_us / T
    # compulsory HBM bytes of the group as launched: every launch reads its source once and
    # writes its (pooled) output once; the first also reads the canvas plane
    enc_bytes = 4.0 * sb['canvas'].numel()
    for st_ in eng.plan['ccnn']:
      src_ = sb['img'] if st_[1] == 0 else sb['ccnn'][st_[1] - 1]
      enc_bytes += 4.0 * (src_.numel() + sb['ccnn'][st_[-1]].numel())
    achieved = tot_f * Bs / (enc_us * 1e-6) / 1e12
    enc_traffic, enc_traffic_src = pmc_traffic(Bs, S, want_source=True)
    out['roofline'] = {
        'kernel': 'ra::cpair::conv_pair8_mfma / conv_pair_persist_mfma + ra::wino::conv_wino_mfma + ra::conv::conv3x3_mfma (controller CNN: %d layers in %d '
                  'launches per timestep per sub-batch of %d images)' % (d['ccnn_nlayers'],
                                                                       len(eng.plan['ccnn']), Bs),
        'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
        'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': enc_traffic, 'traffic_source': enc_traffic_src, 'mfma_busy': mfma_busy(),
        'executed': mfma_executed(enc_us),
        'traffic_note': 'HBM bytes per launch group from committed rocprofv3 --pmc passes '
                        '(profiles/r0x_pmc_encoder_traffic.json; FETCH_SIZE x2 gfx950 correction + '
                        'WRITE_SIZE); null if no pass matches this shape',
        'algorithmic_bytes_per_launch_group': enc_bytes,
        'peak_note': 'dense f32-input MFMA (v_mfma_f32_16x16x4_f32); float32 arithmetic throughout.  achieved = ALGORITHMIC FLOPs (2*9*Cin*Cout per output pixel, SURVEY 8d) / time: the Winograd layers (L4-L6, K1w) execute 2.25x fewer multiplies than that count',
        'flop_per_launch_group': tot_f * Bs, 'avg_us_per_launch_group': enc_us,
        'first_layer_cache': None if 'l0cache' not in sb else {
            'us_per_forward': cache_us,
            'note': 'the image channels\' share of layer 0 (27 of its 36 multiply-adds per output) is '
                    'timestep-invariant (SURVEY.md Appendix A): the first timestep (zero canvas) runs the '
                    'un-cached kernel and writes it as a by-product; us_per_forward = that launch minus a cached '
                    'one, and its 1/T share is included in avg_us_per_launch_group.  flop_per_launch_group stays the ALGORITHMIC '
                    'count (SURVEY 8d), of which %.1f %% is no longer executed per timestep'
                    % (100.0 * 0.75 * per_f[0] / tot_f)},
        'layers': layers}

    if coalesce > 1 and pipe.slots:
      # what the pipeline's slots really launch: the same group over the images of `coalesce` batches (the figures above are per
      # batch of 8 images — the unit of the metric, comparable across rounds, and the shape the committed PMC passes ran)
      reng = pipe.slots[0][0]
      rsb = reng.subs[0]
      rB = int(rsb['img'].shape[0])

      def enc_launched():
        for st_ in reng.plan['ccnn']:
          src_ = rsb['img'] if st_[1] == 0 else rsb['ccnn'][st_[1] - 1]
          reng._run_cnn([st_], reng.W['ccnn'], src_, rsb['ccnn'], 1, 'ctrl_cnn', plane=rsb['canvas'], cache=rsb.get('l0cache'))
      us_l = graph_time_us(enc_launched, tickets=True) + cache_us * rB / Bs / T
      out['roofline']['as_launched'] = {
          'images_per_launch': rB, 'avg_us_per_launch_group': us_l, 'flop_per_launch_group': tot_f * rB,
          'achieved': tot_f * rB / (us_l * 1e-6) / 1e12, 'frac': tot_f * rB / (us_l * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS,
          'note': 'the pipeline slots decode %d batches per forward (config.batches_per_launch): the same launch group over %d images' % (coalesce, rB)}

    Hh = S
    prefilled = not d['disable_overwrite']
    pflags = (ops.PASTE_Y_PREFILLED if prefilled else 0) | ops.PASTE_CANVAS_FLOORED

    def attn_group(bb=sb, n=Bs):  # the two launches of rounds 1-4: extract, paste
      ops.extract_direct(bb['img'][:n], 0, bb['attn'][0][:n], Fh, Fw, d['C0p'], True, bb['x_patch'][0][:n],
                         canvas=bb['canvas'][:n], canvas_chan=d['D'])
      ops.paste_direct(bb['y_out_patch'][0][:n], 0, bb['attn'][0][:n], -5.0, d['disable_overwrite'],
                       bb['y_out'].data_ptr(), T * S * S, S, S, canvas=bb['canvas'][:n], flags=pflags)

    # The forward issues extract and paste as TWO launches (ra_engine.fuse_extract_conv0 is off by default: the fused
    # extract + attention-CNN layer 0 launch, ra_extract_conv0_f32, was built in round 5, is parity-tested and measured SLOWER —
    # profiles/r05_attn_fusion_probe.txt).  Only with RA_FUSE_EXTRACT_CONV0=1 does the accounting below switch to
    #   group = (extract+conv0 launch) + (paste launch) - (the conv0 launch alone)
    # all three measured in the same graphs-of-8 protocol; `unfused` = extract + paste as separate launches = what runs.
    fused_attn = bool(eng.fuse_extract_conv0 and 'acnn0_plain' in Wt and eng.plan['acnn'][0] == ('single', 0))

    def fused_group(bb=sb, n=Bs, W_=None):
      W_ = Wt if W_ is None else W_
      _, sc0, sh0, c0, _ = W_['acnn'][0]
      ops.extract_conv0(bb['img'][:n], 0, bb['attn'][0][:n], Fh, Fw, True, bb['x_patch'][0][:n], W_['acnn0_plain'], sc0[0], sh0[0], c0,
                        True, bb['acnn'][0][:n], canvas=bb['canvas'][:n], canvas_chan=d['D'])
      ops.paste_direct(bb['y_out_patch'][0][:n], 0, bb['attn'][0][:n], -5.0, d['disable_overwrite'],
                       bb['y_out'].data_ptr(), T * S * S, S, S, canvas=bb['canvas'][:n], flags=pflags)

    def conv0_alone(bb=sb, n=Bs, W_=None):
      W_ = Wt if W_ is None else W_
      wp0, sc0, sh0, c0, _ = W_['acnn'][0]
      ops.conv3x3(bb['x_patch'][0][:n], wp0, sc0[0], sh0[0], c0, relu=True, pool=1, out=bb['acnn'][0][:n])

    def attn_times(bb=sb, n=Bs, W_=None):
      """{'unfused': extract + paste, 'fused': extract+conv0 + paste, 'conv0': the absorbed launch alone, 'net': the group's time}"""
      r = {'unfused': graph_time_us(lambda: attn_group(bb, n))}
      if fused_attn:
        r['fused'] = graph_time_us(lambda: fused_group(bb, n, W_))
        r['conv0'] = graph_time_us(lambda: conv0_alone(bb, n, W_))
        r['net'] = r['fused'] - r['conv0']
      else:
        r['net'] = r['unfused']
      return r

    def prefill(bb=sb, ride=rides):  # once per forward: y_out = sigmoid(beta) everywhere, unless that rides on the first
      if not ride:                   # controller-CNN launch (then it is inside roofline.first_layer_cache); the canvas is
        ops.fill(bb['y_out'], 1.0 / (1.0 + np.exp(5.0)))  # zeroed by the input-packing launch

    attn_t = attn_times()
    attn_us = attn_t['net']
    # the kernels are window-only, so their time depends on the attention box: the same group at three box sizes (the
    # headline figures use the bench model's own box, 0.35 of the image side: seed_weights)
    by_box = {}
    rec_keep = sb['attn'][0].clone()
    for frac in (0.15, 0.35, 1.0):
      rec = rec_keep.clone()
      rec[:, 0], rec[:, 1] = 0.5 * S, 0.5 * S                       # centre
      rec[:, 2], rec[:, 3] = frac * S, frac * S                     # size
      rec[:, 4] = rec[:, 5] = float(np.log(frac * S / Fh))          # lg_var = log(size / F) (full_model.py:702-709)
      sb['attn'][0].copy_(rec)
      bt = attn_times()
      us = bt['net']
      by_box['%.2f' % frac] = {'extract_paste_us': us, 'launches': bt,
                               'frac_algorithmic': float(S * S * (d['acnn_channels'][0] + 3) * 4) * Bs / (us * 1e-6) / 1e9 / PEAK_HBM_GBS}
    sb['attn'][0].copy_(rec_keep)
    # the window-only paste relies on once-per-forward fills: their 1/T share belongs to every timestep's
    # attention-resample time
    fill_us = graph_time_us(prefill, reps=10, inner=2) if (prefilled and not rides) else 0.0
    group_us = attn_us + fill_us / T
    attn_bytes = float(S * S * (d['acnn_channels'][0] + 3) * 4) * Bs
    attn_traffic, attn_traffic_src = pmc_traffic(Bs, S, 'rNN_pmc_attn_traffic.json', want_source=True)
    out['roofline_attn'] = {
        'kernel': 'ra::attnd::extract_rows_kernel + paste_win_kernel (+ 1/T of the per-forward fills) — attention '
                  'resample, one sub-batch of %d images' % Bs, 'bound': 'hbm',
        'achieved': attn_bytes / (group_us * 1e-6) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
        'frac': attn_bytes / (group_us * 1e-6) / 1e9 / PEAK_HBM_GBS, 'traffic': attn_traffic, 'traffic_source': attn_traffic_src,
        'achieved_traffic': None if attn_traffic is None else attn_traffic / (group_us * 1e-6) / 1e9,
        'bytes_per_launch_group': attn_bytes, 'avg_us_per_launch_group': group_us,
        'extract_paste_us': attn_us, 'fills_us_per_forward': fill_us,
        'launches_us': attn_t,
        'accounting': ('RA_FUSE_EXTRACT_CONV0=1: the extract runs FUSED with layer 0 of the attention CNN (one launch instead of two); the '
                       "group's time = (extract+conv0 launch + paste launch) - (the conv0 launch alone) = launches_us.fused - "
                       'launches_us.conv0; launches_us.unfused = extract + paste as the two separate launches of rounds 1-4, '
                       'frac_unfused the fraction on that time') if fused_attn else 'extract + paste, two launches',
        'frac_unfused': attn_bytes / ((attn_t['unfused'] + fill_us / T) * 1e-6) / 1e9 / PEAK_HBM_GBS,
        'by_box_size': by_box,
        'frac_window_is_image': by_box['1.00']['frac_algorithmic'], 'frac_small_box_0.15': by_box['0.15']['frac_algorithmic'],
        'y_out_prefill': ('rides on the first timestep\'s first controller-CNN launch (MFMA-bound, HBM idle): its cost is '
                          'inside roofline.first_layer_cache.us_per_forward') if rides else 'its own launch, inside fills_us_per_forward',
        'launch_floor_us': graph_time_us(lambda: ops.fill(sb['attn'][0][:1, :4], 0.0)),
        'note': 'achieved = ALGORITHMIC bytes (SURVEY 8d: read the attention input once, write y_out, '
                'read+write the canvas = H*W*(C0+3)*4 B per image-timestep) / time of the two dependent launches.  The '
                'kernels are window-only: `traffic` (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, the file traffic_source names) is '
                'what they really move and achieved_traffic = traffic / time.  launch_floor_us = one dependent launch of a '
                '4-float fill in the same graph: two of them are the part of the group no kernel design can remove'}
    # the same group at a large batch, where the fixed latencies amortise
    import full_model as fm2
    m32 = fm2.get_model(opt, is_training=False)
    seed_weights(m32, 99)
    m32.engine.forward(torch.rand((32, S, S, 3), generator=g, dtype=torch.float32).cuda())
    s32 = m32.engine.subs[0]
    r32 = m32.engine._prefill_rides(s32)
    t32 = attn_times(s32, 32, m32.engine.W)
    us32 = t32['net']
    f32us = graph_time_us(lambda: prefill(s32, r32), reps=5, inner=1) if (prefilled and not r32) else 0.0
    by32 = float(S * S * (d['acnn_channels'][0] + 3) * 4) * 32
    out['roofline_attn']['at_B32'] = {'avg_us_per_launch_group': us32 + f32us / T, 'extract_paste_us': us32, 'launches_us': t32,
                                      'achieved_algorithmic': by32 / ((us32 + f32us / T) * 1e-6) / 1e9,
                                      'frac_algorithmic': by32 / ((us32 + f32us / T) * 1e-6) / 1e9 / PEAK_HBM_GBS,
                                      'note': 'can exceed 1 by construction: the definition counts whole planes, the kernels '
                                              'touch the attention window only (traffic 0.26x algorithmic); not a roofline claim'}
    del m32, s32
    torch.cuda.empty_cache()
    if coalesce > 1 and pipe.slots:
      # ... and at the size the pipeline's slots launch it (config.batches_per_launch batches per forward)
      reng_a = pipe.slots[0][0]
      rsb_a = reng_a.subs[0]
      rB_a = int(rsb_a['img'].shape[0])
      rsb_a['attn'][0].copy_(sb['attn'][0][:1].expand(rB_a, -1))  # the bench model's box for every image of the slot
      tl = attn_times(rsb_a, rB_a, reng_a.W)
      byl = float(S * S * (d['acnn_channels'][0] + 3) * 4) * rB_a
      out['roofline_attn']['as_launched'] = {
          'images_per_launch': rB_a, 'extract_paste_us': tl['net'], 'frac_algorithmic': byl / (tl['net'] * 1e-6) / 1e9 / PEAK_HBM_GBS,
          'note': 'the same two launches over the %d images of a pipeline slot; algorithmic bytes, as frac (the north star\'s bar '
                  'is quoted at the configuration\'s B = 8: frac)' % rB_a}
    # the three figures side by side at the top level — algorithmic bytes at the configuration's B (the north
    # star's bar: 0.60), the same at the images a pipeline slot launches, and the bytes the counters say really moved
    ra_ = out['roofline_attn']
    ra_['frac_as_launched'] = ra_.get('as_launched', {}).get('frac_algorithmic')
    ra_['frac_traffic'] = None if ra_['achieved_traffic'] is None else ra_['achieved_traffic'] / PEAK_HBM_GBS
    ra_['summary'] = ('frac %.3f at B = %d (north star 0.60: %s); %s at the %s images a pipeline slot launches; real HBM traffic %s of 8 TB/s — the two '
                      'dependent launches are latency-bound, not bandwidth-bound: the bar is met from 16 images per launch upwards'
                      % (ra_['frac'], Bs, 'met' if ra_['frac'] >= 0.6 else 'NOT met',
                         'n/a' if ra_['frac_as_launched'] is None else '%.3f' % ra_['frac_as_launched'],
                         ra_.get('as_launched', {}).get('images_per_launch', 'n/a'),
                         'n/a' if ra_['frac_traffic'] is None else '%.3f' % ra_['frac_traffic']))
    # the whole post-encoder tail of one timestep exactly as the forward issues it
    tail_us = graph_time_us(lambda: eng._launch_tail(sb, 1, False, sb['ccnn'][-1]))
    out['tail_us'] = tail_us
    if 'ctrl_ws' in sb:
      lone_ctrl = ops.controller_batch if sb.get('ctrl_batch') else ops.controller_split  # more than 14 images: the group-shared form
      out['controller_us'] = graph_time_us(lambda: lone_ctrl(
          eng.desc, sb['ccnn'][-1], Wt['ctrl_split'], sb['h_last'][0], sb['ctrl_out'][0],
          sb['gmaps'][0], sb['attn'][0], sb['ctrl_ws'], sb['ctrl_status']))
      out['controller_status'] = int(sb['ctrl_status'].item())
      if pipe.slots and pipe.slots[0][0].subs[0].get('ctrl_batch'):  # the form the pipeline's slots run
        ps = pipe.slots[0][0].subs[0]
        out['controller_us_in_slots'] = graph_time_us(lambda: ops.controller_batch(
            eng.desc, ps['ccnn'][-1], pipe.slots[0][0].W['ctrl_split'], ps['h_last'][0], ps['ctrl_out'][0], ps['gmaps'][0],
            ps['attn'][0], ps['ctrl_ws'], ps['ctrl_status']))
        out['controller_in_slots'] = 'group-shared (K2b: 16 workgroups per %d images)' % ops_group(pipe.slots[0][0])
    else:
      out['controller_us'] = graph_time_us(lambda: ops.controller(
          eng.desc, sb['ccnn'][-1], Wt['ctrl'], sb['h_last'][0], sb['ctrl_out'][0], sb['gmaps'][0],
          sb['attn'][0]))
    # the Hungarian op is not roofline-rated; report us per [T,T] problem (device
    # solver, one wave per problem, B problems per launch) on matching-shaped weights
    rng = np.random.RandomState(7)
    wm = np.zeros((B, T, T), np.float32)
    for bb in range(B):
      k = rng.randint(T // 2, T)
      for i, j in enumerate(rng.permutation(k)):
        wm[bb, i, j] = rng.uniform(0.5, 0.95)
      wm[bb, :k, :k] += rng.uniform(0, 0.05, (k, k)).astype(np.float32)
    wm_d = torch.as_tensor(np.floor(wm * 1e6 + 0.5) / 1e6 + 1e-5).cuda()
    ops.hungarian(wm_d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
      ops.hungarian(wm_d)
    torch.cuda.synchronize()
    out['hungarian_us_per_problem'] = 1e6 * (time.perf_counter() - t0) / (5 * B)
    out['hungarian_us_per_problem_note'] = 'toy matrices (a planted permutation + 5 % noise), B problems per launch; the ' \
        'representative figure is hungarian_cfg4_step'
    # ... and on the matrices a real cfg4 training step produces (tests/golden/hungarian_cfg4_step.npz: the 2 B = 16 mask + box
    # problems of one step of a freshly initialised model, near-uniform IoUs — the regime the 1e-6 quantisation exists for)
    fx = os.path.join(ROOT, 'tests', 'golden', 'hungarian_cfg4_step.npz')
    if os.path.exists(fx):
      step_w = torch.as_tensor(np.load(fx)['weights']).cuda()
      n_p = int(step_w.shape[0])

      def hung_us(w, reps=5):
        ops.hungarian(w)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
          ops.hungarian(w)
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t1) / reps
      launch = hung_us(step_w)
      single = [hung_us(step_w[k:k + 1].contiguous(), 3) for k in range(n_p)]
      fxz = np.load(fx)
      host_us = None
      if 'iou' in fxz and 's_gt' in fxz:  # the same problems through the training step's default path: host cores, a stream host function
        h_iou, h_s = torch.as_tensor(fxz['iou']).cuda(), torch.as_tensor(fxz['s_gt']).cuda()
        _, _, blk = ops.segm_match_host(h_iou, h_s, None, threads=min(32, _cgroup_cpus()))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
          ops.segm_match_host(h_iou, h_s, blk, threads=min(32, _cgroup_cpus()))
        torch.cuda.synchronize()
        host_us = 1e6 * (time.perf_counter() - t1) / 5
      out['hungarian_cfg4_step'] = {
          'host_node_us': host_us,
          'host_node_note': 'ra_segm_match_host_f32 (round 6, the training step\'s default): precondition kernel, D2H, ra_hungarian_f32 on host '
                            'threads side by side as a host function of the stream, H2D, re-mask — the whole f_segm_match of the %d problems; '
                            'launch_us is the device solver alone' % n_p,
          'problems': n_p, 'shape': list(step_w.shape[1:]), 'launch_us': launch, 'us_per_problem_in_launch': launch / n_p,
          'us_per_problem_alone_median': float(np.median(single)), 'us_per_problem_alone_max': float(np.max(single)),
          'note': 'one wave per problem, all problems of the step in ONE launch (its duration = the slowest problem, '
                  'which is what the training step waits for); alone = one problem per launch'}
    out['config']['sub_batches'] = len(eng.subs)
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline(opt, 1234)
    print(json.dumps(out))
  if world > 1:
    ra_dist.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
  main()
