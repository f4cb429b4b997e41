"""Debug aid (GPU): inside the training graph, every ConvBNActPool call is evaluated by the HIP
function AND by a torch float32 stand-in on the same inputs (forward values and input / weight
gradients for a random upstream gradient); prints the calls that disagree."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('rec-attend-public_amd', 'oracle', 'tests'):
  sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import full_model, ra_train
import test_train_gpu as tt
from train_debug6 import conv_ref as _cr  # noqa
import train_debug3
conv_ref = train_debug3.conv_ref
ra_train.ConvBNActPool = train_debug3.HIP[0]

HIP = ra_train.ConvBNActPool
calls = []

def both(x, w, b, gamma, beta, meta):
  y, mean, var = HIP.apply(x, w, b, gamma, beta, meta)
  with torch.enable_grad():
    xs = [t.detach().clone().requires_grad_(True) for t in (x, w, b, gamma, beta)]
    xh = [t.detach().clone().requires_grad_(True) for t in (x, w, b, gamma, beta)]
    yr, mr, vr = conv_ref(*xs, meta)
    yh, mh, vh = HIP.apply(*xh, meta)
    g = torch.randn_like(yr)
    (yr * g).sum().backward()
    (yh * g).sum().backward()
  rel = lambda a, b_: float((a - b_).abs().max() / max(float(b_.abs().max()), 1e-6))
  rec = {'shape': tuple(x.shape), 'w': tuple(w.shape), 'tr': meta['transposed'], 'stride': meta['stride'], 'pool': meta['pool'],
         'contig': x.is_contiguous(), 'ingraph_vs_clone': rel(y, yh), 'ptrs': [t.data_ptr() % 256 for t in (x, w, b, gamma, beta)], 'y': rel(yh, yr), 'mean': rel(mh, mr), 'var': rel(vh, vr),
         'dx': rel(xh[0].grad, xs[0].grad), 'dw': rel(xh[1].grad, xs[1].grad), 'dgamma': rel(xh[3].grad, xs[3].grad),
         'dbeta': rel(xh[4].grad, xs[4].grad)}
  calls.append(rec)
  return y, mean, var

class FakeApply:
  def __init__(self, fn): self.apply = fn
ra_train.ConvBNActPool = FakeApply(both)
opt, P, x, y_gt, s_gt = tt._case(T=2)
m = full_model.get_model(opt).load_weights(P)
ts = ra_train.TrainStep(m)
loss, pieces, st = ts.forward_loss(x, y_gt, s_gt)
for i, r in enumerate(calls):
  worst = max(r[k] for k in ('y', 'mean', 'var', 'dx', 'dw', 'dgamma', 'dbeta', 'ingraph_vs_clone'))
  print('%2d %s %s' % (i, 'BAD' if worst > 1e-3 else 'ok ', {k: (('%.1e' % v) if isinstance(v, float) else v) for k, v in r.items()}))
