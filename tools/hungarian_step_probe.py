#!/usr/bin/env python
"""The Hungarian launch of the training bench's step, problem by problem: captures the [2 B, T, T] matrices the merged
f_segm_match of one cfg4-shaped step solves (B mask problems, then B box problems) and times the device solver on all of
them, on each half and on every single problem (us, median of 5)."""
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd'))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import ra_native as rn
if os.environ.get('RA_LIB'):  # A/B: another build of the library (same ABI)
  rn.LIB_PATH = os.environ['RA_LIB']
import bench
import full_model
import full_model_train as fmt
import ra_ops as ops

opt = bench.make_opt('cvppp', 512, 512, 16)
opt.update(use_knob=True, knob_base=1.0, knob_decay=0.9, steps_per_knob_decay=300, knob_box_offset=300, knob_segm_offset=500,
           knob_use_timescale=True, gt_box_ctr_noise=0.05, gt_box_pad_noise=0.1, gt_segm_noise=0.3, base_learn_rate=1e-3,
           learn_rate_decay=0.96, steps_per_learn_rate_decay=5000, seed=1234)   # bench.train_steps' options
m = full_model.get_model(opt, is_training=True)
rng = np.random.RandomState(1234)
x, y_gt, s_gt = fmt.synthetic_batch(rng, 8, 512, 512, 16)
got = {}
orig = ops.segm_match


def spy(iou, s):
  got['iou'], got['s'] = iou.clone(), s.clone()
  return orig(iou, s)


ops.segm_match = spy
m.run(['loss', 'train_step'], {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': False})
ops.segm_match = orig
iou, s = got['iou'], got['s']
print('problems', tuple(iou.shape), 'live ground-truth columns per problem', s.sum(1).cpu().numpy().astype(int).tolist())


def t(i, sg):
  ts = []
  for _ in range(5):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(i, sg)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
  return float(np.median(ts))


B2 = iou.shape[0]
print('all %d: %.0f us   masks: %.0f us   boxes: %.0f us' % (B2, t(iou, s), t(iou[:B2 // 2].contiguous(), s[:B2 // 2].contiguous()),
                                                           t(iou[B2 // 2:].contiguous(), s[B2 // 2:].contiguous())))
print('single problems (us):', [round(t(iou[k:k + 1].contiguous(), s[k:k + 1].contiguous())) for k in range(B2)])
np.save(os.path.join(ROOT, 'gpurun_out', 'hung_iou.npy'), iou.cpu().numpy())
np.save(os.path.join(ROOT, 'gpurun_out', 'hung_s.npy'), s.cpu().numpy())
