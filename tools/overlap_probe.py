#!/usr/bin/env python
"""Does the latency-bound tail of a timestep (extract, 13 patch convs, score, paste) overlap with a
full-chip controller-CNN pass when both are in flight at once?  Each chain is captured as its own
LINEAR HIP graph (8 repetitions, submitted in one shot) and replayed on its own stream; compare the
concurrent wall time with the two serial times.  (Feasibility probe for DESIGN.md §9-1.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd'))
import torch
import bench, full_model

opt = bench.make_opt('cvppp', 512, 512, 16)
model = full_model.get_model(opt, is_training=False)
bench.seed_weights(model, 1234)
eng = model.engine
x = torch.rand((8, 512, 512, 3), generator=torch.Generator().manual_seed(1)).cuda()
eng.forward(x)
torch.cuda.synchronize()
sb = eng.subs[0]
REP = 8


NSTEPS = int(sys.argv[1]) if len(sys.argv) > 1 else len(eng.plan['ccnn'])  # encoder launches in the probe


def enc():
  eng._run_cnn(eng.plan['ccnn'][:NSTEPS], eng.W['ccnn'], sb['img'], sb['ccnn'], 1, 'ctrl_cnn', plane=sb.get('canvas'))


def tail():
  # everything of _launch_tail after the controller, for timestep 0 (its attention record is valid)
  eng._launch_tail_after_ctrl(sb, 0) if hasattr(eng, '_launch_tail_after_ctrl') else None


# build the tail without the controller by temporarily stubbing it
import ra_ops as ops
real_split, real_ctrl = ops.controller_split, ops.controller
def tail():
  ops.controller_split = lambda *a, **k: None
  ops.controller = lambda *a, **k: None
  try:
    eng._launch_tail(sb, 0, False, sb['ccnn'][-1])
  finally:
    ops.controller_split, ops.controller = real_split, real_ctrl

sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
graphs = {}
for name, fn, st in (('enc', enc, sB), ('tail', tail, sA)):
  with torch.cuda.stream(st):
    fn()
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g, stream=st):
    for _ in range(REP):
      fn()
  graphs[name] = (g, st)


def run(names, reps=10):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(reps):
    for n in names:
      g, st = graphs[n]
      with torch.cuda.stream(st):
        g.replay()
  torch.cuda.synchronize()
  return 1e6 * (time.perf_counter() - t0) / (reps * REP)

for _ in range(2):
  e, t, both = run(['enc']), run(['tail']), run(['tail', 'enc'])
  print('per repetition: encoder alone %.1f us, tail alone %.1f us, both in flight %.1f us (sum %.1f, max %.1f)'
        % (e, t, both, e + t, max(e, t)))
