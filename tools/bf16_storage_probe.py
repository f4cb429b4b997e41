import sys, os
sys.path.insert(0, 'rec-attend-public_amd'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np, torch
import ra_oracle as ora, ra_oracle_torch as ort, ra_train, full_model
src = open('tests/test_train_gpu.py').read()
ns = {'__name__': 'x'}
exec(compile(src.split('def test_conv_layer_forward_backward')[0].replace('pytestmark = pytest.mark.gpu', ''), 'x', 'exec'), ns)
def _grad_cosine(gref, got_of, P, wd):
  dots = np.zeros(3)
  for k, g in gref.items():
    if not ('_cnn_b_' in k or '_dcnn_b_' in k):
      got = got_of(k) + (wd * P[k] if ra_train.is_decayed(k) else 0.0)
      dots += [float((got * g).sum()), float((got * got).sum()), float((g * g).sum())]
  return dots[0] / np.sqrt(dots[1] * dots[2])
ns['_grad_cosine'] = _grad_cosine
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2
opt, P, x, y_gt, s_gt = ns['_case'](wmul=0.6, T=T)
orc = {}
for kind in (None, 'bf16', 'bf16s'):
  ort.set_conv_operands(kind)
  try:
    head, g, st = ns['_oracle_grads'](opt, P, x, y_gt, s_gt)
  finally:
    ort.set_conv_operands(None)
  orc[kind] = (float(head['loss']), g, st)
  print('oracle', kind, 'loss', orc[kind][0])
wd = float(opt['weight_decay'])
for store in ('0', '1'):
  os.environ['RA_BF16_STORE'] = store
  m = full_model.get_model(dict(opt, compute_dtype='bf16')).load_weights(P)
  ts = ra_train.TrainStep(m); ts.seq_ctrl_split = False
  ts.bucket.zero_grad()
  loss, pieces, st = ts.forward_loss(x, y_gt, s_gt)
  loss.backward(); torch.cuda.synchronize()
  got_of = lambda k: ts.bucket.grad_of[k].cpu().numpy()
  print('product store=%s loss %.5f' % (store, float(loss)), ' cosine vs oracle f64 %.4f  bf16 %.4f  bf16s %.4f' % tuple(
      ns['_grad_cosine'](orc[k][1], got_of, P, wd) for k in (None, 'bf16', 'bf16s')))
  for kind in ('bf16', 'bf16s'):
    dev = max(ns['_rel'](st[k][0].cpu().numpy(), orc[kind][2][k][0].numpy()) for k in st if k.startswith('ctrl_cnn_') and k.endswith('_0'))
    print('    ctrl-CNN t=0 statistics vs oracle %s: worst mean deviation %.2e' % (kind, dev))
