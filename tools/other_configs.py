#!/usr/bin/env python
"""Throughput of the decode loop on BASELINE.json's other configurations (parity-test cases, not
bench lines): cfg3 KITTI arch 128x448 T=20 B=16 and cfg5 Cityscapes arch 256x512 T=20 B=4 per GPU,
synthetic inputs and random-init weights like bench.py."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd'))
import numpy as np
import torch
import bench, full_model


def opt_of(arch, H, W, T):
  o = bench.make_opt('cvppp', H, W, T)
  if arch == 'cvppp':
    return o
  o.update(ctrl_cnn_depth=[16, 16, 32, 32, 64, 64, 64, 64], ctrl_cnn_pool=[2, 2, 1, 2, 1, 2, 1, 2],
           attn_cnn_depth=[16, 32, 32, 64, 64, 96], attn_dcnn_depth=[64, 64, 32, 32, 16, 16, 1],
           dynamic_var=True, fixed_gamma=False, add_skip_conn=True, add_d_out=True, add_y_out=True,
           attn_add_d_out=True, attn_add_y_out=True, ctrl_add_d_out=True, ctrl_add_y_out=True,
           attn_cnn_skip='1,0,1,0,1,0,1,0')  # run_kitti.sh:68-111
  if arch == 'cityscapes':
    o.update(num_semantic_classes=9, fixed_gamma=True, use_iou_box=True)  # run_cityscapes.sh:62-110
  return o


for name, arch, H, W, T, B in (('cfg3', 'kitti', 128, 448, 20, 16), ('cfg5', 'cityscapes', 256, 512, 20, 4),
                               ('cfg2', 'cvppp', 512, 512, 16, 8)):
  opt = opt_of(arch, H, W, T)
  m = full_model.get_model(opt, is_training=False)
  bench.seed_weights(m, 1)
  g = torch.Generator().manual_seed(2)
  feed = {'x': torch.rand((B, H, W, 3), generator=g).cuda()}
  if opt['add_d_out']:
    nc = opt['num_semantic_classes']
    feed['d_in'] = torch.nn.functional.one_hot(torch.randint(0, 8, (B, H, W), generator=g), 8).float().cuda()
    feed['y_in'] = torch.softmax(torch.randn((B, H, W, nc), generator=g), dim=-1).cuda()
  eng = m.engine
  for _ in range(3):
    eng.forward(feed['x'], d_in=feed.get('d_in'), y_in=feed.get('y_in'))
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  n = 10
  for _ in range(n):
    eng.forward(feed['x'], d_in=feed.get('d_in'), y_in=feed.get('y_in'))
  torch.cuda.synchronize()
  ms = 1e3 * (time.perf_counter() - t0) / n
  print('%s %-10s %dx%d T=%d B=%d: %.2f ms per forward, %.0f instance-timesteps/s' % (name, arch, H, W, T, B, ms, B * T / ms * 1e3))
