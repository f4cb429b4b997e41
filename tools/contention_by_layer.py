#!/usr/bin/env python
"""Per launch of the controller CNN: its time alone and with controllers (or the whole tail) of another slot as company, with the
static tile walk and with drawn tiles (ra_tile_tickets_bind).  usage: contention_by_layer.py [images per slot = 16]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import torch
import bench, full_model
import ra_ops as ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T, S, REP = 16, 512, 16
opt = bench.make_opt('cvppp', S, S, T)
engs = []
for k in range(2):
  m = full_model.get_model(opt, is_training=False)
  bench.seed_weights(m, 1234 + k)
  m.engine.co_resident = 4
  m.engine.forward(torch.rand((B, S, S, 3)).cuda())
  engs.append(m.engine)
torch.cuda.synchronize()
e0, e1 = engs
sb = e0.subs[0]


def capture(fn):
  fn(); torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    fn()
  return g


def layer_graph(step, tickets):
  def run():
    bound = tickets and ops.tickets_bind(sb['tickets'])
    try:
      src = sb['img'] if step[1] == 0 else sb['ccnn'][step[1] - 1]
      for _ in range(REP):
        e0._run_cnn([step], e0.W['ccnn'], src, sb['ccnn'], 1, 'ctrl_cnn', plane=sb['canvas'], cache=sb.get('l0cache'))
    finally:
      if bound:
        ops.tickets_unbind()
  return capture(run)


def company_graph(kind):
  s1 = e1.subs[0]
  def run():
    for _ in range(2):
      for tt in range(T):
        if kind == 'ctrl':
          (ops.controller_batch if s1.get('ctrl_batch') else ops.controller_split)(
              e1.desc, s1['ccnn'][-1], e1.W['ctrl_split'], s1['h_last'][tt], s1['ctrl_out'][tt], s1['gmaps'][tt], s1['attn'][tt],
              s1['ctrl_ws'], s1['ctrl_status'])
        else:
          e1._launch_tail(s1, tt, False, s1['ccnn'][-1])
  return capture(run)


sa, sc = torch.cuda.Stream(), torch.cuda.Stream()


def timed(g, company=None, n=4):
  a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  if company is not None:
    with torch.cuda.stream(sc):
      for _ in range(2 * n):
        company.replay()
  with torch.cuda.stream(sa):
    g.replay()
    a0.record()
    for _ in range(n):
      g.replay()
    a1.record()
  torch.cuda.synchronize()
  return 1e3 * a0.elapsed_time(a1) / (n * REP)


comp = {k: company_graph(k) for k in ('ctrl', 'tail')}
print('images per slot %d; us per launch: alone | with controllers | with the whole tail of another slot' % B)
for step in e0.plan['ccnn']:
  row = []
  for tickets in (False, True):
    g = layer_graph(step, tickets)
    row.append((timed(g), timed(g, comp['ctrl']), timed(g, comp['tail'])))
  print('layers %-8s static %6.1f %6.1f %6.1f   drawn %6.1f %6.1f %6.1f' % ((str(list(step[1:])),) + row[0] + row[1]))
