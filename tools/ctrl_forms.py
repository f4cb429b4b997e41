#!/usr/bin/env python
"""us per launch of the three controller forms at cfg2 (B = 8) and cfg3 (B = 16) shapes."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import torch
import bench, full_model, ra_ops as ops
def t_us(fn, reps=30):
  fn(); torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(8):
      fn()
  g.replay(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    g.replay()
  e1.record(); torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / (reps * 8)
for arch, H, W, B in (('cvppp', 512, 512, 8), ('kitti', 128, 448, 16), ('kitti', 128, 448, 8)):
  opt = bench.make_opt(arch, H, W, 4)
  m = full_model.get_model(opt, is_training=False)
  bench.seed_weights(m, 1)
  e = m.engine
  e.prepare(torch.device('cuda'))
  d, Wt = m.dims, e.W
  feat = torch.rand(B, d['G'], d['ccnn_channels'][-1], device='cuda')
  z = lambda *s: torch.zeros(s, device='cuda')
  h, co, gm, at = z(B, d['hid']), z(B, 9), z(B, d['iters'], d['G']), z(B, 16)
  out = ['%s %dx%d B=%d:' % (arch, H, W, B)]
  out.append('one-workgroup %.1f' % t_us(lambda: ops.controller(e.desc, feat, Wt['ctrl'], h, co, gm, at)))
  if B <= 14:
    ws, st = ops.ctrl_split_workspace(e.desc, B, 'cuda')
    out.append('split %.1f' % t_us(lambda: ops.controller_split(e.desc, feat, Wt['ctrl_split'], h, co, gm, at, ws, st)))
  if ops.ctrl_batch_supported(e.desc):
    ws, st = ops.ctrl_batch_workspace(e.desc, B, 'cuda')
    out.append('group-shared %.1f (status %d)' % (t_us(lambda: ops.controller_batch(e.desc, feat, Wt['ctrl_split'], h, co, gm, at, ws, st)), int(st.item())))
  print(' | '.join(out))
