#!/usr/bin/env python
"""Controller kernel timing vs iteration count (marginal cost per iteration / fixed cost)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import numpy as np, torch
import ra_ops as ops, ra_oracle as ora

def graph_time_us(fn, reps=50):
  fn(); torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(8): fn()
  for _ in range(3): g.replay()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(reps): g.replay()
  e1.record(); torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / reps / 8

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for iters, ng in ((1, 2), (2, 2), (3, 2), (5, 2), (5, 1)):
  opt = ora.make_opt('cvppp', 512, 512, 2, num_ctrl_rnn_iter=iters, num_glimpse_mlp_layers=ng)
  d = ora.derive(opt); P = ora.random_params(opt, 1)
  desc = ops.make_ctrl_desc(d['G'], 64, 256, iters, ng, 1, 256, 512, 512, 48, 48, 0, 0, 0, 1)
  lstm = {k[len('ctrl_lstm_'):]: v for k, v in P.items() if k.startswith('ctrl_lstm_')}
  gmw = [(P['glimpse_mlp_w_%d' % i], P['glimpse_mlp_b_%d' % i]) for i in range(ng)]
  cmw = [(P['ctrl_mlp_w_0'], P['ctrl_mlp_b_0'])]
  dev = lambda a: torch.from_numpy(a).cuda()
  wp2 = dev(ops.pack_ctrl_split_weights(desc, lstm, gmw, cmw)); wp1 = dev(ops.pack_ctrl_weights(desc, lstm, gmw, cmw))
  ws, st = ops.ctrl_split_workspace(desc, B, 'cuda')
  feat = torch.rand(B, d['G'], 64).cuda()
  z = lambda *s: torch.zeros(s, device='cuda')
  h, co, gm, at = z(B, 256), z(B, 9), z(B, iters, d['G']), z(B, 16)
  t2 = graph_time_us(lambda: ops.controller_split(desc, feat, wp2, h, co, gm, at, ws, st))
  t1 = graph_time_us(lambda: ops.controller(desc, feat, wp1, h, co, gm, at))
  print('B=%d iters=%d n_gmlp=%d  split %.1f us   single %.1f us   status %d' % (B, iters, ng, t2, t1, int(st.item())))
