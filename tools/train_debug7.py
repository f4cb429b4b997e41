"""Debug aid (GPU): controller-CNN features of timestep 0 and 1 (HIP function / torch float32
stand-in) against the float64 oracle's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('rec-attend-public_amd', 'oracle', 'tests', 'tools'):
  sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import torch.nn.functional as F
import full_model, ra_train
import ra_oracle as ora, ra_oracle_torch as ort
import test_train_gpu as tt
from train_debug6 import conv_ref  # noqa

class FakeApply:
  def __init__(self, fn): self.apply = fn
HIP, REF = ra_train.ConvBNActPool, FakeApply(conv_ref)
opt, P, x, y_gt, s_gt = tt._case(T=2)
# oracle features: rerun the oracle forward capturing the ctrl-CNN output per timestep
feats64 = []
orig_cnn = ort.cnn
def cap(xx, PP, scope, n, pools, tt_, use_bn):
  hs = orig_cnn(xx, PP, scope, n, pools, tt_, use_bn)
  if scope == 'ctrl_cnn': feats64.append([h.detach().numpy() for h in hs])
  return hs
ort.cnn = cap
fwd, _ = ort.forward(opt, P, x, phase_train=True)
ort.cnn = orig_cnn
for name, impl in (('hip', HIP), ('ref', REF)):
  ra_train.ConvBNActPool = impl
  m = full_model.get_model(opt).load_weights(P)
  ts = ra_train.TrainStep(m)
  got = []
  o_cnn = ts._cnn
  def cnn(xx, scope, *a, **k):
    hs = o_cnn(xx, scope, *a, **k)
    if scope == 'ctrl_cnn': got.append([h.detach().cpu().numpy() for h in hs])
    return hs
  ts._cnn = cnn
  with torch.no_grad():
    loss, pieces, st = ts.forward_loss(x, y_gt, s_gt)
  for t in range(2):
    print(name, 't=%d' % t, ['%.1e' % (np.abs(a - b).max() / np.abs(b).max()) for a, b in zip(got[t], feats64[t])])
