#!/usr/bin/env python
"""Does a concurrently running controller slow the controller CNN down?  (round 5, the pipeline's 14 % above its encoder floor)
Stream A replays a graph of one forward's 16 controller-CNN launch groups; stream B meanwhile replays a graph holding only
controllers (or only the patch-sized tail without the controller).  A's time, from HIP events on A, alone and under each kind
of company.  usage: contention_probe.py [images per slot = 16]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import torch
import bench, full_model
import ra_ops as ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T, S = 16, 512
opt = bench.make_opt('cvppp', S, S, T)
engs = []
for k in range(2):
  m = full_model.get_model(opt, is_training=False)
  bench.seed_weights(m, 1234 + k)
  m.engine.co_resident = 4
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
  bound = e.tile_tickets and ops.tickets_bind(sb['tickets'])  # as the engine's forward does (RA_ENGINE_TICKETS=0: the static walk)
  try:
    for tt in range(T):
      e._launch_encoder(sb, max(tt, 1))
  finally:
    if bound:
      ops.tickets_unbind()


def ctrl_only(e, reps=3):
  sb = e.subs[0]
  for _ in range(reps):
    for tt in range(T):
      (ops.controller_batch if sb.get('ctrl_batch') else ops.controller_split)(
          e.desc, sb['ccnn'][-1], e.W['ctrl_split'], sb['h_last'][tt], sb['ctrl_out'][tt], sb['gmaps'][tt], sb['attn'][tt],
          sb['ctrl_ws'], sb['ctrl_status'])


def tail_no_ctrl(e, reps=3):
  sb = e.subs[0]
  saved = {}
  for n in ('controller', 'controller_split', 'controller_batch'):
    saved[n] = getattr(ops, n)
    setattr(ops, n, lambda *a, **k: None)
  try:
    for _ in range(reps):
      for tt in range(T):
        e._launch_tail(sb, tt, False, sb['ccnn'][-1])
  finally:
    for n, f in saved.items():
      setattr(ops, n, f)


def tail_full(e, reps=2):
  sb = e.subs[0]
  for _ in range(reps):
    for tt in range(T):
      e._launch_tail(sb, tt, False, sb['ccnn'][-1])


gE = capture(lambda: enc_only(engs[0]))
sa, sb_ = torch.cuda.Stream(), torch.cuda.Stream()


def time_graph(g, s, n=5):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  with torch.cuda.stream(s):
    g.replay()
    e0.record()
    for _ in range(n):
      g.replay()
    e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n


def timed(company):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  n = 4
  with torch.cuda.stream(sb_):
    f0.record()
    for _ in range(3 * n):
      company.replay()
    f1.record()
  with torch.cuda.stream(sa):
    e0.record()
    for _ in range(n):
      gE.replay()
    e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / n, f0.elapsed_time(f1) / (3 * n)


tE = time_graph(gE, sa)
print('images per slot %d; controller form: %s; tile tickets %s' % (B, 'group-shared' if engs[0].subs[0].get('ctrl_batch') else 'per-image split', 'on' if engs[0].tile_tickets else 'off'))
print('controller CNN of one forward (16 groups) alone: %.3f ms  = %.1f us per group' % (tE, 1e3 * tE / T))
for name, fn in (('controllers only', ctrl_only), ('tail without controller', tail_no_ctrl), ('whole tail', tail_full)):
  gC = capture(lambda: fn(engs[1]))
  tC = time_graph(gC, sb_)
  a, c = timed(gC)
  print('%-26s company alone %.3f ms; together: controller CNN %.3f ms (x%.3f), company %.3f ms (x%.3f); sum of alone %.3f, max %.3f'
        % (name, tC, a, a / tE, c, c / tC, tE + tC, max(tE, tC)))
