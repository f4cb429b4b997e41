#!/usr/bin/env python
"""Timing of the loss / statistics head at cfg2 shapes (B=8, T=16, 512x512): the streaming
pair_stats pass (HBM-bound: reads a and b once), get_gt_box, the matching and the scalar tail."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd'))
import torch
import ra_ops as ops

B, T, S = 8, 16, 512
g = torch.Generator().manual_seed(0)
a = torch.rand((B, T, S, S), generator=g).cuda()
b = (torch.rand((B, T, S, S), generator=g) > 0.9).float().cuda()
s_gt = torch.ones((B, T)).cuda()


def timeit(fn, reps=20):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record(); torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / reps


us = timeit(lambda: ops.pair_stats(a, b))
byts = 2.0 * B * T * S * S * 4
print('pair_stats      %8.1f us  %6.0f GB/s (%.0f%% of 8 TB/s) on %.0f MB' % (us, byts / us / 1e3, byts / us / 1e3 / 80, byts / 1e6))
us = timeit(lambda: ops.gt_box(b, 0.2, 20.0))
print('gt_box          %8.1f us  %6.0f GB/s (read + write %.0f MB)' % (us, byts / us / 1e3, byts / 1e6))
st = ops.pair_stats(a, b)
us = timeit(lambda: ops.segm_match(st['iou_soft'], s_gt))
print('segm_match      %8.1f us  (B=%d problems of %dx%d)' % (us, B, T, T))
m, _ = ops.segm_match(st['iou_soft'], s_gt)
us = timeit(lambda: ops.loss_stats(st['iou_soft'], st['iou_hard'], st['dice_hard'], m, st['iou_soft'], m, s_gt, s_gt, st['sum_b']))
print('loss_stats      %8.1f us' % us)
# realistic matching problems: IoU of noisy predictions of distinct ellipses
import numpy as np
import time
rng = np.random.RandomState(0)
iou = np.zeros((B, T, T), np.float32)
for bb in range(B):
  k = rng.randint(8, T)
  perm = rng.permutation(k)
  for i in range(k):
    iou[bb, i, perm[i]] = rng.uniform(0.5, 0.95)
  iou[bb, :k, :k] += rng.uniform(0, 0.05, (k, k)).astype(np.float32)
  s_gt[bb, k:] = 0
d_iou = torch.as_tensor(iou).cuda()
us = timeit(lambda: ops.segm_match(d_iou, s_gt))
print('segm_match      %8.1f us  (realistic: one dominant match per instance + noise)' % us)
w = np.ascontiguousarray(st['iou_soft'].cpu().numpy())
t0 = time.perf_counter(); ops.hungarian(w); t1 = time.perf_counter()
print('host hungarian on the adversarial (near-uniform) problem set: %.1f us' % (1e6 * (t1 - t0)))
t0 = time.perf_counter(); ops.hungarian(iou); t1 = time.perf_counter()
print('host hungarian on the realistic set: %.1f us' % (1e6 * (t1 - t0)))
