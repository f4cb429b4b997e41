#!/usr/bin/env python
"""What bounds the decode pipeline?  Per-batch time of graphs holding only the controller CNN
(16 launch groups), only the rest of the timestep (16 tails), or both, replayed on 1..4 streams."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import torch
import bench, full_model
K, B, T, S = 4, 8, 16, 512
opt = bench.make_opt('cvppp', S, S, T)
engs = []
for k in range(K):
  m = full_model.get_model(opt, is_training=False)
  bench.seed_weights(m, 1234 + k)
  m.engine.forward(torch.rand((B, S, S, 3)).cuda())
  engs.append(m.engine)
torch.cuda.synchronize()

def capture(fn):
  fn(); torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    fn()
  return g

def enc_only(e):
  sb = e.subs[0]
  for tt in range(T):
    e._launch_encoder(sb, max(tt, 1))
def tail_only(e):
  sb = e.subs[0]
  for tt in range(T):
    e._launch_tail(sb, tt, False, sb['ccnn'][-1])
def both(e):
  sb = e.subs[0]
  for tt in range(T):
    e._launch_tail(sb, tt, False, e._launch_encoder(sb, max(tt, 1)))

for name, fn in (('controller CNN only', enc_only), ('tail only', tail_only), ('both', both)):
  graphs = [capture(lambda e=e: fn(e)) for e in engs]
  for n in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(n)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = 12
    for _ in range(steps):
      for g, s in zip(graphs[:n], streams):
        with torch.cuda.stream(s):
          g.replay()
    torch.cuda.synchronize()
    print('%-20s %d streams: %.3f ms per batch' % (name, n, 1e3 * (time.perf_counter() - t0) / (steps * n)))
