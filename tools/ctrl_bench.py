#!/usr/bin/env python
"""Controller kernel timing vs iteration count (marginal cost per glimpse / fixed cost), split
(16 workgroups per image) against single-workgroup form.  Usage: ctrl_bench.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd'))
import torch
import bench, full_model
import ra_ops as ops


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
  opt = bench.make_opt('cvppp', 512, 512, 2)
  opt.update(num_ctrl_rnn_iter=iters, num_glimpse_mlp_layers=ng)
  m = full_model.get_model(opt, is_training=False)
  bench.seed_weights(m, 1)
  eng = m.engine
  eng.forward(torch.rand((B, 512, 512, 3)).cuda())
  torch.cuda.synchronize()
  sb, d = eng.subs[0], eng.d
  t2 = graph_time_us(lambda: ops.controller_split(eng.desc, sb['ccnn'][-1], eng.W['ctrl_split'], sb['h_last'][0],
                                                  sb['ctrl_out'][0], sb['gmaps'][0], sb['attn'][0], sb['ctrl_ws'],
                                                  sb['ctrl_status']))
  t1 = graph_time_us(lambda: ops.controller(eng.desc, sb['ccnn'][-1], eng.W['ctrl'], sb['h_last'][0],
                                            sb['ctrl_out'][0], sb['gmaps'][0], sb['attn'][0]))
  print('B=%d iters=%d n_gmlp=%d  split %.1f us   single %.1f us   status %d'
        % (B, iters, ng, t2, t1, int(sb['ctrl_status'].item())))
