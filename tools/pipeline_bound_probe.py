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
  m.engine.co_resident = K  # as a DecodePipeline slot: the group-shared controller
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

import ra_ops as ops
def both_minus(skip):
  """the whole timestep with one kind of tail launch made a no-op (results are garbage; timing only)"""
  def run(e):
    saved = {}
    names = {'controller': ['controller', 'controller_split', 'controller_batch'], 'attn': ['extract_direct', 'paste_direct', 'paste_score_direct'],
             'patch': ['conv3x3', 'conv_pair']}[skip]
    if skip == 'patch':  # only the patch-sized convs: wrap and filter on the input size
      o3, op = ops.conv3x3, ops.conv_pair
      ops.conv3x3 = lambda x, *a, **k: (k.get('out') if x.shape[1] <= 48 and k.get('out') is not None else o3(x, *a, **k))
      ops.conv_pair = lambda x, *a, **k: (k.get('out') if x.shape[1] <= 48 and k.get('out') is not None else op(x, *a, **k))
      try:
        both(e)
      finally:
        ops.conv3x3, ops.conv_pair = o3, op
      return
    for n in names:
      saved[n] = getattr(ops, n)
      setattr(ops, n, lambda *a, **k: None)
    try:
      both(e)
    finally:
      for n, f in saved.items():
        setattr(ops, n, f)
  return run

for name, fn in (('controller CNN only', enc_only), ('tail only', tail_only), ('both', both),
                 ('both - controller', both_minus('controller')), ('both - extract/paste', both_minus('attn')),
                 ('both - patch convs', both_minus('patch'))):
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
