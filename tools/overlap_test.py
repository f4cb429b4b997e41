#!/usr/bin/env python
"""Do two independent latency-bound kernels on two HIP streams overlap (eager vs captured graph)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import torch, ra_ops as ops, ra_oracle as ora

opt = ora.make_opt('cvppp', 512, 512, 2)
d = ora.derive(opt); P = ora.random_params(opt, 1)
desc = ops.make_ctrl_desc(d['G'], 64, 256, 5, 2, 1, 256, 512, 512, 48, 48, 0, 0, 0, 1)
lstm = {k[len('ctrl_lstm_'):]: v for k, v in P.items() if k.startswith('ctrl_lstm_')}
gmw = [(P['glimpse_mlp_w_%d' % i], P['glimpse_mlp_b_%d' % i]) for i in range(2)]
cmw = [(P['ctrl_mlp_w_0'], P['ctrl_mlp_b_0'])]
wp = torch.from_numpy(ops.pack_ctrl_split_weights(desc, lstm, gmw, cmw)).cuda()
B = 4
def mk():
  ws, st = ops.ctrl_split_workspace(desc, B, 'cuda')
  z = lambda *s: torch.zeros(s, device='cuda')
  return dict(feat=torch.rand(B, d['G'], 64).cuda(), h=z(B, 256), co=z(B, 9), gm=z(B, 5, d['G']), at=z(B, 16), ws=ws, st=st)
A, Bb = mk(), mk()
run = lambda s: ops.controller_split(desc, s['feat'], wp, s['h'], s['co'], s['gm'], s['at'], s['ws'], s['st'])
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def serial():
  for _ in range(4): run(A); run(Bb)
def parallel():
  main = torch.cuda.current_stream()
  s1.wait_stream(main); s2.wait_stream(main)
  with torch.cuda.stream(s1):
    for _ in range(4): run(A)
  with torch.cuda.stream(s2):
    for _ in range(4): run(Bb)
  main.wait_stream(s1); main.wait_stream(s2)

def time_eager(fn, reps=20):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): fn()
  e1.record(); torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / reps
def time_graph(fn, reps=20):
  fn(); torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g): fn()
  g.replay(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps): g.replay()
  e1.record(); torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / reps
print('eager  serial %.1f us   2-stream %.1f us' % (time_eager(serial), time_eager(parallel)))
print('graph  serial %.1f us   2-stream %.1f us' % (time_graph(serial), time_graph(parallel)))
