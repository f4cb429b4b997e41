"""Debug aid (GPU): swap the HIP conv function for the torch stand-in per sub-network."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('rec-attend-public_amd', 'oracle', 'tests', 'tools'):
  sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import full_model, ra_train
import test_train_gpu as tt
from train_debug3 import conv_ref, FakeApply  # noqa (runs debug3 once on import; fine)

import train_debug3
HIP = train_debug3.HIP[0]
REF = FakeApply(conv_ref)
T = int(os.environ.get('T', '2'))
opt, P, x, y_gt, s_gt = tt._case(T=T)
head, gref, stats = tt._oracle_grads(opt, P, x, y_gt, s_gt)
wd = float(opt['weight_decay'])
for mode in ('none', 'ctrl_cnn', 'attn_cnn', 'attn_dcnn', 'ctrl_cnn+attn_cnn', 'attn_cnn+attn_dcnn', 'all'):
  m = full_model.get_model(opt).load_weights(P)
  ts = ra_train.TrainStep(m)
  o_cnn, o_dcnn = ts._cnn, ts._dcnn
  def cnn(xx, scope, *a, **k):
    ra_train.ConvBNActPool = REF if (scope in mode or mode == 'all') else HIP
    return o_cnn(xx, scope, *a, **k)
  def dcnn(xx, scope, *a, **k):
    ra_train.ConvBNActPool = REF if (scope in mode or mode == 'all') else HIP
    return o_dcnn(xx, scope, *a, **k)
  ts._cnn, ts._dcnn = cnn, dcnn
  ts.bucket.zero_grad()
  loss, pieces, st = ts.forward_loss(x, y_gt, s_gt)
  loss.backward()
  g = {k: ts.bucket.grad_of[k].cpu().numpy() + (wd * P[k] if ra_train.is_decayed(k) else 0) for k in gref}
  errs = sorted(((float(np.abs(g[k] - gref[k]).max() / max(np.abs(gref[k]).max(), 1e-3)), k) for k in gref if '_b_' not in k), reverse=True)
  print('torch stand-in for [%s]: loss %.6f  worst:' % (mode, float(loss)), [(k, '%.3f' % e) for e, k in errs[:3]], 'match', pieces['match'].flatten().tolist(), 'mbox', pieces['match_box'].flatten().tolist(), 'oracle', head['match'].flatten().tolist(), head['match_box'].flatten().tolist())
