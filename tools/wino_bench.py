#!/usr/bin/env python
"""Winograd vs direct conv at the controller CNN's mid layers (cfg2, B = 8): us per launch."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import numpy as np, torch
import ra_ops as ops
def t_us(fn, reps=50):
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
rng = np.random.RandomState(0)
for name, H, Ci, Co, pool in (('L4', 128, 16, 32, 1), ('L5', 128, 32, 32, 2), ('L6', 64, 32, 64, 2)):
  x = torch.randn(8, H, H, Ci, device='cuda')
  w = (rng.randn(3, 3, Ci, Co) / np.sqrt(9 * Ci)).astype(np.float32)
  sc, sh = [torch.from_numpy(a).cuda() for a in ops.fold_bn(None, Co)]
  wd, ww = torch.from_numpy(ops.pack_conv_weights(w)).cuda(), torch.from_numpy(ops.pack_wino_weights(w)).cuda()
  yo = torch.empty(8, H // pool, H // pool, Co, device='cuda')
  d = t_us(lambda: ops.conv3x3(x, wd, sc, sh, Co, relu=True, pool=pool, out=yo))
  wi = t_us(lambda: ops.conv_wino(x, ww, sc, sh, Co, relu=True, pool=pool, out=yo))
  gf = 2 * 9 * Ci * Co * H * H * 8 / 1e9
  print('%s %dx%d %d->%d pool %d: direct %.1f us (%.0f TF/s), winograd %.1f us (%.0f TF/s algorithmic)' % (
      name, H, H, Ci, Co, pool, d, gf / d * 1e-3 * 1e3 / 1e3 * 1e3 if False else gf / (d * 1e-6) / 1e3, wi, gf / (wi * 1e-6) / 1e3))
