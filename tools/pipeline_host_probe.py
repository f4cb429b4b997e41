#!/usr/bin/env python
"""Host time of the decode pipeline's loop: how long submit() takes when a slot is free (the GPU idle: pure host cost), and the
share of a steady-state step the host spends inside submit / retire rather than waiting.  usage: pipeline_host_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import torch
import bench, full_model
B, T, S = 8, 16, 512
opt = bench.make_opt('cvppp', S, S, T)
m = full_model.get_model(opt, is_training=False)
bench.seed_weights(m, 1234)
feed = {'x': torch.rand((B, S, S, 3)).cuda(), 'phase_train': False}
pipe = m.pipeline(4, coalesce=2)
for _ in range(24):
  while pipe.full():
    pipe.retire()
  pipe.submit(['y_out', 's_out'], feed)
pipe.drain()
torch.cuda.synchronize()
# (a) submits into an empty pipeline: the host cost alone
t = []
for _ in range(8):
  t0 = time.perf_counter()
  pipe.submit(['y_out', 's_out'], feed)
  t.append(1e3 * (time.perf_counter() - t0))
pipe.drain()
print('submit() into free slots, ms each (every second one launches a forward of two batches): ' + ' '.join('%.3f' % v for v in t))
# (b) steady state
N = 200
ts = tr = 0.0
t00 = time.perf_counter()
for _ in range(N):
  t0 = time.perf_counter()
  while pipe.full():
    pipe.retire()
  t1 = time.perf_counter()
  pipe.submit(['y_out', 's_out'], feed)
  t2 = time.perf_counter()
  tr += t1 - t0
  ts += t2 - t1
pipe.drain()
tot = time.perf_counter() - t00
print('steady state: %.3f ms per batch; in submit %.3f ms, in retire (waiting for the oldest batch + collecting) %.3f ms' % (1e3 * tot / N, 1e3 * ts / N, 1e3 * tr / N))
