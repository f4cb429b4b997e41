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


if __name__ == '__main__':
  make('full_model_cvppp_128', 'cvppp', 128, 128, 5, 1, 16)   # BASELINE.json configs[0]
  make('full_model_kitti_64x96', 'kitti', 64, 96, 3, 2, 41)   # KITTI-style flags
