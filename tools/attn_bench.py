#!/usr/bin/env python
"""Graph-replay timing of the attention kernels at cfg2 shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd')); sys.path.insert(0, ROOT)
import torch, bench, full_model, ra_ops as ops

def gtime(fn, reps=50):
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
S, T = 512, 2
m = full_model.get_model(bench.make_opt('cvppp', S, S, T)); bench.seed_weights(m, 1)
eng = m.engine; eng.nsub = 1; eng.use_graph = False
eng.forward(torch.rand(B, S, S, 3).cuda()); torch.cuda.synchronize()
sb, d = eng.subs[0], eng.d
print('attn rec', sb['attn'][0][0, :6].tolist())
ex = lambda: ops.extract_direct(sb['img'], 0, sb['attn'][0], 48, 48, 4, True, sb['x_patch'][0], canvas=sb['canvas'], canvas_chan=3)
import os
PF = int(os.environ.get('PASTE_FLAGS', '3'))
pa = lambda: ops.paste_direct(sb['y_out_patch'][0], 0, sb['attn'][0], -5.0, False, sb['y_out'].data_ptr(), T * S * S, S, S, canvas=sb['canvas'], flags=PF)
print('B=%d extract_direct %.1f us   paste_direct %.1f us  (%.0f GB/s on 12 B/px)' % (B, gtime(ex), gtime(pa), B * S * S * 12 / gtime(pa) / 1e3))
rec0 = sb['attn'][0].clone()
for name, mod in (('box off-image', lambda r: r.__setitem__((slice(None), 0), -5000.0)),
                  ('box = whole image', lambda r: (r.__setitem__((slice(None), slice(2, 4)), 500.0), r.__setitem__((slice(None), slice(4, 6)), 2.34))),
                  ('small box 60px', lambda r: (r.__setitem__((slice(None), slice(2, 4)), 60.0), r.__setitem__((slice(None), slice(4, 6)), 0.22)))):
  sb['attn'][0].copy_(rec0); mod(sb['attn'][0])
  print('%-18s extract %.1f us  paste %.1f us' % (name, gtime(ex), gtime(pa)))
for cy in (-300.0, -1000.0, -5000.0, 5000.0, 900.0):
  sb['attn'][0].copy_(rec0); sb['attn'][0][:, 0] = cy
  print('ctr_y %8.1f  paste %.1f us' % (cy, gtime(pa, reps=10)))
sb['attn'][0].copy_(rec0); sb['attn'][0][:, 1] = -5000.0
print('ctr_x -5000  paste %.1f us' % gtime(pa, reps=10))
