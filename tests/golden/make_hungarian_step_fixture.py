"""Writes tests/golden/hungarian_cfg4_step.npz: the [2 B, T, T] soft-IoU matrices (B mask problems, then B box problems) and
the s_gt rows that the merged f_segm_match of ONE cfg4-shaped training step hands to the Hungarian op (modellib.py:382-415;
512 x 512, T = 16, B = 8, the bench's synthetic batch and a freshly initialised model: near-uniform IoUs, the regime the
reference's 1e-6 quantisation exists for), together with the matching the plain-C oracle (oracle/hungarian_oracle.c) returns
for them through the reference's pre-conditioning.

The matrices are captured on an MI355X by tools/hungarian_step_probe.py (-> gpurun_out/hung_iou.npy, hung_s.npy); this
script only adds the oracle's answer and runs on the CPU:  python tests/golden/make_hungarian_step_fixture.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ra_oracle as ora  # noqa: E402

iou = np.load(os.path.join(ROOT, 'gpurun_out', 'hung_iou.npy')).astype(np.float32)
s = np.load(os.path.join(ROOT, 'gpurun_out', 'hung_s.npy')).astype(np.float32)
w, mask_x, mask_y = ora.f_segm_match_precondition(iou, s)
match = ora.hungarian_c(w) * mask_x * mask_y
np.savez_compressed(os.path.join(HERE, 'hungarian_cfg4_step.npz'), iou=iou, s_gt=s, weights=w, match=match.astype(np.float32))
print('problems', iou.shape, 'live columns', s.sum(1).astype(int).tolist(), 'matched', match.sum((1, 2)).astype(int).tolist())
