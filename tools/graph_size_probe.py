#!/usr/bin/env python
"""Does a timestep cost more inside a large HIP graph?  Graphs holding n = 1, 2, 4, 8, 16 whole timesteps (controller CNN + tail)
of one cfg2 forward, replayed back to back: us per timestep by HIP events on the stream, and by wall clock for ONE sync-bracketed
replay (what config.lone_batch_ms measures).  usage: graph_size_probe.py [B = 8]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import torch
import bench, full_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T, S = 16, 512
opt = bench.make_opt('cvppp', S, S, T)
m = full_model.get_model(opt, is_training=False)
bench.seed_weights(m, 1234)
e = m.engine
e.forward(torch.rand((B, S, S, 3)).cuda())
torch.cuda.synchronize()
sb = e.subs[0]


def steps(n):
  for tt in range(n):
    e._launch_tail(sb, tt % T, False, e._launch_encoder(sb, max(tt % T, 1)))


def capture(fn):
  fn(); torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    fn()
  return g


print('B = %d; us per timestep' % B)
for n in (1, 2, 4, 8, 16, 32):
  g = capture(lambda: steps(n))
  reps = max(2, 64 // n)
  for _ in range(2):
    g.replay()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  e0.record()
  for _ in range(reps):
    g.replay()
  e1.record()
  torch.cuda.synchronize()
  ev = 1e3 * e0.elapsed_time(e1) / (reps * n)
  lone = []
  for _ in range(11):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    lone.append(1e6 * (time.perf_counter() - t0))
  lone.sort()
  t0 = time.perf_counter()
  for _ in range(5):
    g.replay()
  host = 1e6 * (time.perf_counter() - t0) / 5
  torch.cuda.synchronize()
  print('  graph of %2d timesteps (%3d replays back to back): %6.1f us by events;  one sync-bracketed replay: %7.1f us = %6.1f per timestep;  host time of a replay call %6.1f us'
        % (n, reps, ev, lone[len(lone) // 2], lone[len(lone) // 2] / n, host))
