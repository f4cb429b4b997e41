#!/usr/bin/env python
"""Graph-replay timing of the attention kernels at cfg2 shapes (the bench's attention records).
  RA_PASTE_GEO=<rows>,<threads> picks the paste geometry (read once per process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd')); sys.path.insert(0, ROOT)
import torch, bench, full_model, ra_ops as ops

def gtime(fn, reps=50, inner=8):
  fn(); torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(inner): fn()
  for _ in range(3): g.replay()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(reps): g.replay()
  e1.record(); torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / reps / inner

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S, T = 512, 2
m = full_model.get_model(bench.make_opt('cvppp', S, S, T)); bench.seed_weights(m, 1)
eng = m.engine; eng.nsub = 1; eng.use_graph = False
eng.forward(torch.rand(B, S, S, 3).cuda()); torch.cuda.synchronize()
sb, d = eng.subs[0], eng.d
tag = 'GEO=%s' % os.environ.get('RA_PASTE_GEO', '-')
print(tag, 'attn rec', [round(v, 2) for v in sb['attn'][0][0, :9].tolist()])
tiny = torch.zeros(64, device='cuda')
print('floor: 1-element fill launch %.2f us' % gtime(lambda: ops.fill(tiny, 1.0)))
ex = lambda: ops.extract_direct(sb['img'], 0, sb['attn'][0], 48, 48, 4, True, sb['x_patch'][0], canvas=sb['canvas'], canvas_chan=3)
def pa(flags):
  return lambda: ops.paste_direct(sb['y_out_patch'][0], 0, sb['attn'][0], -5.0, False, sb['y_out'].data_ptr(), T * S * S, S, S, canvas=sb['canvas'], flags=flags)
te = gtime(ex)
for fl in (3, 2, 0):
  tp = gtime(pa(fl))
  both = gtime(lambda: (ex(), pa(fl)()))
  print('%s B=%d flags=%d extract %.2f us  paste %.2f us  pair %.2f us -> %.0f GB/s algorithmic (%.3f of 8 TB/s)'
        % (tag, B, fl, te, tp, both, B * S * S * 28 / both / 1e3, B * S * S * 28 / both / 1e3 / 8000))
rec0 = sb['attn'][0].clone()
for name, mod in (('box = whole image', lambda r: (r.__setitem__((slice(None), slice(2, 4)), 500.0), r.__setitem__((slice(None), slice(4, 6)), 2.34))),
                  ('small box 60px', lambda r: (r.__setitem__((slice(None), slice(2, 4)), 60.0), r.__setitem__((slice(None), slice(4, 6)), 0.22)))):
  sb['attn'][0].copy_(rec0); mod(sb['attn'][0])
  print('%s %-18s extract %.2f us  paste(flags 2) %.2f us' % (tag, name, gtime(ex), gtime(pa(2))))
