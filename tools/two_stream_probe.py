#!/usr/bin/env python
"""Probe: do two whole-batch forward graphs replayed on two HIP streams overlap (the tail of one
under the controller CNN of the other)?  Prints ms per batch for 1 stream and for K streams."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import torch
import bench, full_model
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, T, S = 8, 16, 512
opt = bench.make_opt('cvppp', S, S, T)
engs, xs = [], []
for k in range(K):
  m = full_model.get_model(opt, is_training=False)
  bench.seed_weights(m, 1234 + k)
  engs.append(m.engine)
  xs.append(torch.rand((B, S, S, 3)).cuda())
for e, x in zip(engs, xs):
  e.forward(x); e.forward(x)
torch.cuda.synchronize()
def timeit(n_eng, steps=10):
  streams = [torch.cuda.Stream() for _ in range(n_eng)]
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    for e, x, s in zip(engs[:n_eng], xs, streams):
      with torch.cuda.stream(s):
        e.forward(x)
  torch.cuda.synchronize()
  return 1e3 * (time.perf_counter() - t0) / (steps * n_eng)
for rep in range(2):
  for n in range(1, K + 1):
    print('%d streams : %.3f ms per batch' % (n, timeit(n, 20)))

def timeit_bounded(n_eng, limit, steps=40, snap=False):
  """host blocks on the oldest batch's event once `limit` batches are pending"""
  streams = [torch.cuda.Stream() for _ in range(n_eng)]
  torch.cuda.synchronize()
  pend, keep = [], []
  t0 = time.perf_counter()
  for i in range(steps):
    k = i % n_eng
    if len(pend) >= limit:
      pend.pop(0).synchronize()
      if keep: keep.pop(0)
    with torch.cuda.stream(streams[k]):
      engs[k].forward(xs[k])
      if snap:
        keep.append(engs[k].fetch('y_out').clone())
      ev = torch.cuda.Event(); ev.record(streams[k])
    pend.append(ev)
  torch.cuda.synchronize()
  return 1e3 * (time.perf_counter() - t0) / steps
for n in (3, 4):
  for limit in (n, n + 1, n + 2, 2 * n):
    print('%d streams, <=%d pending: %.3f ms per batch; with y_out snapshot %.3f' % (
        n, limit, timeit_bounded(n, limit), timeit_bounded(n, limit, snap=True)))
