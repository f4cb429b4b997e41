#!/usr/bin/env python
"""Writes the golden full_model fixtures from the NumPy oracle (seeded inputs + weights ->
expected outputs).  The reference has no fixtures for the neural path and cannot run here
(TF 0.12), so these vectors pin the ORACLE (regression) and give the GPU tests data that does
not depend on re-running it.  Weights are regenerated from the seed by ra_oracle.random_params."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(HERE)), 'oracle'))
import ra_oracle as ora  # noqa: E402


def make(name, arch, H, W, T, B, seed):
  opt = ora.make_opt(arch, H, W, T)
  P = ora.random_params(opt, seed)
  rng = np.random.RandomState(seed + 1)
  x = rng.rand(B, H, W, 3).astype(np.float32)
  kw = {}
  if opt['add_d_out']:
    kw['d_in'] = np.eye(8, dtype=np.float32)[rng.randint(0, 8, (B, H, W))]
    kw['y_in'] = ora.softmax(rng.randn(B, H, W, opt['num_semantic_classes'])).astype(np.float32)
  r = ora.full_model_forward(opt, P, x, **kw)
  out = {k: r[k].astype(np.float32) for k in ('y_out', 's_out', 'attn_ctr', 'attn_size', 'x_patch',
                                               'ctrl_rnn_glimpse_map')}
  np.savez_compressed(os.path.join(HERE, name + '.npz'), opt=np.array(opt, dtype=object),
                      seed=seed, x=x, **kw, **out)
  print(name, 'max y', float(r['y_out'].max()), 'frac>0.5', float((r['y_out'] > 0.5).mean()))


def synth_gt(rng, B, T, H, W, max_inst):
  """CVPPP-shaped ground truth: a few ellipses per image sorted by area."""
  yy, xx = np.mgrid[0:H, 0:W]
  y_gt = np.zeros((B, T, H, W), np.float32)
  s_gt = np.zeros((B, T), np.float32)
  for b in range(B):
    inst = []
    for _ in range(rng.randint(1, max_inst + 1)):
      cy, cx = rng.uniform(0.15, 0.85) * H, rng.uniform(0.15, 0.85) * W
      ry, rx = rng.uniform(0.05, 0.2) * H, rng.uniform(0.05, 0.2) * W
      inst.append((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0).astype(np.float32))
    inst.sort(key=lambda m: -m.sum())
    for t, m in enumerate(inst):
      y_gt[b, t], s_gt[b, t] = m, 1.0
  return y_gt, s_gt


def make_loss(name, arch, H, W, T, B, seed):
  """Loss / statistics head (full_model.py:913-1097) on the decode of seeded inputs: the fixture
  holds x, the ground truth (bit-packed) and every scalar + both matchings."""
  opt = ora.make_opt(arch, H, W, T)
  P = ora.random_params(opt, seed)
  rng = np.random.RandomState(seed + 2)
  x = rng.rand(B, H, W, 3).astype(np.float32)
  y_gt, s_gt = synth_gt(rng, B, T, H, W, T - 1)
  out = ora.loss_head(opt, ora.full_model_forward(opt, P, x), y_gt, s_gt)
  keep = {k: np.float64(v) for k, v in out.items() if np.ndim(v) == 0}
  np.savez_compressed(os.path.join(HERE, name + '.npz'), opt=np.array(opt, dtype=object), seed=seed,
                      x=x, y_gt_bits=np.packbits(y_gt.astype(np.uint8)), y_gt_shape=np.array(y_gt.shape),
                      s_gt=s_gt, match=out['match'].astype(np.float32),
                      match_box=out['match_box'].astype(np.float32), **keep)
  print(name, {k: round(float(v), 4) for k, v in keep.items()})


if __name__ == '__main__':
  make_loss('loss_head_cvppp_128', 'cvppp', 128, 128, 5, 2, 16)
  make('full_model_cvppp_128', 'cvppp', 128, 128, 5, 1, 16)   # BASELINE.json configs[0]
  make('full_model_kitti_64x96', 'kitti', 64, 96, 3, 2, 41)   # KITTI-style flags
