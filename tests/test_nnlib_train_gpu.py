"""The nnlib operator surface with phase_train = True (nnlib.py:65-128 batch moments + EMA, :229-253 cnn, :362-400 dcnn,
:476-493 mlp, :637-649 lstm) through the product's closures, against the float64 torch oracle's train-mode operators
(oracle/ra_oracle_torch.py cnn / dcnn / bn_train / mlp / lstm): per-layer lists, batch statistics, EMA shadows, input and
parameter gradients."""
import numpy as np
import pytest
import torch

import ra_oracle_torch as ort

pytestmark = pytest.mark.gpu


def _rel(a, b):
  a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
  return float(np.abs(a - b).max() / max(1e-7, np.abs(b).max()))


def _t64(a, grad=True):
  return torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=grad)


def _oracle_train(fn):
  """Run an oracle operator with BatchNorm on batch statistics, recording them."""
  stats = {}
  ort._BN['train'], ort._BN['stats'] = True, stats
  try:
    return fn(), stats
  finally:
    ort._BN['train'], ort._BN['stats'] = False, None


def _set_bn(model, scope, nlayers, chans, ncopy, rng):
  """Non-trivial gamma / beta and NON-ZERO shadows (so that the 0.9 / 0.1 mix is pinned, not just 0.1 * value)."""
  vals = {}
  for ii in range(nlayers):
    for cp in range(ncopy):
      key = '%s_%d_%d' % (scope, ii, cp)
      c = chans[ii + 1]
      vals[key + '_gamma'] = rng.uniform(0.5, 1.5, c).astype(np.float32)
      vals[key + '_beta'] = (rng.randn(c) * 0.2).astype(np.float32)
      vals[key + '_ema_mean'] = (rng.randn(c) * 0.3).astype(np.float32)
      vals[key + '_ema_var'] = rng.uniform(0.5, 2.0, c).astype(np.float32)
  for k, v in vals.items():
    model[k].copy_(torch.from_numpy(v))
  return vals


@pytest.mark.parametrize('cin,ch,pool,B,H,W', [(3, [8, 8, 16], [1, 2, 2], 2, 24, 32), (4, [16, 32], [2, 1], 3, 16, 16),
                                                (13, [16, 16], [2, 2], 2, 16, 48)])
def test_cnn_train_mode_vs_oracle(cuda, cin, ch, pool, B, H, W):
  import nnlib as nn
  rng = np.random.RandomState(5)
  n = len(ch)
  chans = [cin] + ch
  model = {}
  phase = {'value': True}
  run = nn.cnn([3] * n, chans, pool, [nn.relu] * n, [True] * n, phase_train=phase, wd=1e-4, scope='tcnn', model=model)
  run.declare_copies(2)
  P = {}
  for ii in range(n):
    P['tcnn_w_%d' % ii] = (rng.randn(3, 3, chans[ii], chans[ii + 1]) * 0.25).astype(np.float32)
    P['tcnn_b_%d' % ii] = (rng.randn(chans[ii + 1]) * 0.1).astype(np.float32)
    model['tcnn_w_%d' % ii].copy_(torch.from_numpy(P['tcnn_w_%d' % ii]))
    model['tcnn_b_%d' % ii].copy_(torch.from_numpy(P['tcnn_b_%d' % ii]))
  P.update(_set_bn(model, 'tcnn', n, chans, 2, rng))
  x = rng.randn(B, H, W, cin).astype(np.float32)
  for cp in range(2):  # two calls: the copy counter moves, each copy's shadows take its own call's statistics
    keys = [k for k in P if not k.endswith(('_ema_mean', '_ema_var'))]
    Pt = {k: _t64(P[k], k in keys) for k in P}
    xr = _t64(x)
    hs_r, stats = _oracle_train(lambda: ort.cnn(xr, Pt, 'tcnn', n, pool, cp, True))
    cot = [rng.randn(*h.shape).astype(np.float32) for h in hs_r]
    sum((h * torch.tensor(c, dtype=torch.float64)).sum() for h, c in zip(hs_r, cot)).backward()
    for k in model:
      model[k].grad = None
      model[k].requires_grad_(not k.endswith(('_ema_mean', '_ema_var')))
    xd = torch.tensor(x, device=cuda, requires_grad=True)
    hs = run(xd)
    assert len(hs) == n
    sum((h * torch.tensor(c, device=cuda)).sum() for h, c in zip(hs, cot)).backward()
    for ii in range(n):
      assert tuple(hs[ii].shape) == tuple(hs_r[ii].shape)
      assert _rel(hs[ii].detach().cpu().numpy(), hs_r[ii].detach().numpy()) < 2e-4, (cp, ii)
      key = 'tcnn_%d_%d' % (ii, cp)
      m_r, v_r = stats[key]
      m_d, v_d = run.batch_stats[key]
      assert _rel(m_d.cpu().numpy(), m_r.numpy()) < 1e-4 and _rel(v_d.cpu().numpy(), v_r.numpy()) < 1e-4, key
      # shadow = 0.9 shadow + 0.1 value (nnlib.py:103-110)
      assert _rel(model[key + '_ema_mean'].detach().cpu().numpy(), 0.9 * P[key + '_ema_mean'] + 0.1 * m_r.numpy()) < 1e-5
      assert _rel(model[key + '_ema_var'].detach().cpu().numpy(), 0.9 * P[key + '_ema_var'] + 0.1 * v_r.numpy()) < 1e-5
      other = 'tcnn_%d_%d' % (ii, 1 - cp)
      if cp == 0:  # the other copy's shadows are untouched by this call
        assert (model[other + '_ema_mean'].detach().cpu().numpy() == P[other + '_ema_mean']).all()
    assert _rel(xd.grad.cpu().numpy(), xr.grad.numpy()) < 2e-3, cp
    for k in keys:
      if '_%d_' % (1 - cp) in k and ('gamma' in k or 'beta' in k):
        continue  # the other copy's BN parameters take no part in this call
      got, ref = model[k].grad.cpu().numpy(), Pt[k].grad.numpy()
      if '_b_' in k:  # a bias in front of BatchNorm: zero gradient, both are round-off
        assert np.abs(got - ref).max() < 2e-3, k
      else:
        assert _rel(got, ref) < 3e-3, (cp, k)
  # third call without a copy index: the counter has moved to 2, which declare_copies(2) did not create -> it is created
  with torch.no_grad():
    phase['value'] = False
    run.reset_copy()
    e = run(torch.tensor(x, device=cuda))  # eval mode on the moved shadows: the fused eval kernels
  Pe = {k: _t64(model[k].detach().cpu().numpy(), False) for k in model}
  he = ort.cnn(_t64(x, False), Pe, 'tcnn', n, pool, 0, True)
  assert _rel(e[-1].cpu().numpy(), he[-1].numpy()) < 2e-4


def test_dcnn_train_mode_vs_oracle(cuda):
  """Transposed-conv decoder, stride 2 and 1, with skip connections (concat(prev, skip), nnlib.py:362-366)."""
  import nnlib as nn
  rng = np.random.RandomState(7)
  ch, unpool, skip_ch = [16, 16, 8, 1], [2, 1, 2], [None, 8, 5]
  n = 3
  model = {}
  run = nn.dcnn([3] * n, ch, unpool, [nn.relu] * n, [True] * n, skip_ch=skip_ch, phase_train=True, wd=None, scope='tdc',
                model=model)
  run.declare_copies(1)
  P = {}
  for ii in range(n):
    P['tdc_w_%d' % ii] = (rng.randn(3, 3, ch[ii + 1], run.in_chs[ii]) * 0.25).astype(np.float32)
    P['tdc_b_%d' % ii] = (rng.randn(ch[ii + 1]) * 0.1).astype(np.float32)
    model['tdc_w_%d' % ii].copy_(torch.from_numpy(P['tdc_w_%d' % ii]))
    model['tdc_b_%d' % ii].copy_(torch.from_numpy(P['tdc_b_%d' % ii]))
  P.update(_set_bn(model, 'tdc', n, ch, 1, rng))
  B, H, W = 2, 6, 10
  x = rng.randn(B, H, W, ch[0]).astype(np.float32)
  skips = [None, rng.randn(B, 2 * H, 2 * W, 8).astype(np.float32), rng.randn(B, 2 * H, 2 * W, 5).astype(np.float32)]
  keys = [k for k in P if not k.endswith(('_ema_mean', '_ema_var'))]
  Pt = {k: _t64(P[k], k in keys) for k in P}
  xr = _t64(x)
  sk_r = [None if s is None else _t64(s) for s in skips]
  hs_r, stats = _oracle_train(lambda: ort.dcnn(xr, Pt, 'tdc', n, unpool, 0, sk_r, True))
  cot = [rng.randn(*h.shape).astype(np.float32) for h in hs_r]
  sum((h * torch.tensor(c, dtype=torch.float64)).sum() for h, c in zip(hs_r, cot)).backward()
  for k in model:
    model[k].requires_grad_(not k.endswith(('_ema_mean', '_ema_var')))
  xd = torch.tensor(x, device=cuda, requires_grad=True)
  sk_d = [None if s is None else torch.tensor(s, device=cuda, requires_grad=True) for s in skips]
  hs = run(xd, skip=sk_d)
  sum((h * torch.tensor(c, device=cuda)).sum() for h, c in zip(hs, cot)).backward()
  for ii in range(n):
    assert tuple(hs[ii].shape) == tuple(hs_r[ii].shape)
    assert _rel(hs[ii].detach().cpu().numpy(), hs_r[ii].detach().numpy()) < 2e-4, ii
    key = 'tdc_%d_0' % ii
    assert _rel(run.batch_stats[key][0].cpu().numpy(), stats[key][0].numpy()) < 1e-4
    assert _rel(run.batch_stats[key][1].cpu().numpy(), stats[key][1].numpy()) < 1e-4
    assert _rel(model[key + '_ema_var'].detach().cpu().numpy(), 0.9 * P[key + '_ema_var'] + 0.1 * stats[key][1].numpy()) < 1e-5
  assert _rel(xd.grad.cpu().numpy(), xr.grad.numpy()) < 2e-3
  for i in (1, 2):
    assert _rel(sk_d[i].grad.cpu().numpy(), sk_r[i].grad.numpy()) < 2e-3, i
  for k in keys:
    got, ref = model[k].grad.cpu().numpy(), Pt[k].grad.numpy()
    if '_b_' in k:
      assert np.abs(got - ref).max() < 2e-3, k
    else:
      assert _rel(got, ref) < 3e-3, k


@pytest.mark.parametrize('C', [8, 5, 64])
def test_batch_norm_train_mode_vs_oracle(cuda, C):
  import nnlib as nn
  rng = np.random.RandomState(C)
  x = (rng.randn(3, 10, 14, C) * rng.uniform(0.5, 3.0, C) + rng.randn(C) * 2).astype(np.float32)
  model = {}
  gam, bet = rng.uniform(0.5, 1.5, C).astype(np.float32), (rng.randn(C) * 0.3).astype(np.float32)
  xd = torch.tensor(x, device=cuda, requires_grad=True)
  y = nn.batch_norm(xd, C, True, scope2='tbn', init_beta=bet, init_gamma=gam, model=model)
  P = {'tbn_gamma': _t64(gam), 'tbn_beta': _t64(bet)}
  xr = _t64(x)
  stats = {}
  yr = ort.bn_train(xr, P, 'tbn', stats)
  assert _rel(y.detach().cpu().numpy(), yr.detach().numpy()) < 1e-4
  # shadows start at 0 (tf.train.ExponentialMovingAverage over tensors): one call leaves 0.1 * value
  assert _rel(model['tbn_ema_mean'].cpu().numpy(), 0.1 * stats['tbn'][0].numpy()) < 1e-4
  assert _rel(model['tbn_ema_var'].cpu().numpy(), 0.1 * stats['tbn'][1].numpy()) < 1e-4
  cot = rng.randn(*x.shape).astype(np.float32)
  model['tbn_gamma'].requires_grad_(True)
  model['tbn_beta'].requires_grad_(True)
  y2 = nn.batch_norm(xd, C, True, scope2='tbn', model=model)  # the registered tensors are found again
  (y2 * torch.tensor(cot, device=cuda)).sum().backward()
  (yr * torch.tensor(cot, dtype=torch.float64)).sum().backward()
  assert _rel(xd.grad.cpu().numpy(), xr.grad.numpy()) < 1e-3
  assert _rel(model['tbn_gamma'].grad.cpu().numpy(), P['tbn_gamma'].grad.numpy()) < 1e-3
  assert _rel(model['tbn_beta'].grad.cpu().numpy(), P['tbn_beta'].grad.numpy()) < 1e-3
  assert _rel(model['tbn_ema_mean'].cpu().numpy(), 0.19 * stats['tbn'][0].numpy()) < 1e-4  # 0.9 * 0.1 + 0.1
  # eval mode reads the shadows
  ye = nn.batch_norm(xd.detach(), C, False, scope2='tbn', model=model)
  Pe = {'tbn_' + k: _t64(model['tbn_' + k].detach().cpu().numpy(), False) for k in ('gamma', 'beta', 'ema_mean', 'ema_var')}
  assert _rel(ye.cpu().numpy(), ort.bn_eval(_t64(x, False), Pe, 'tbn').numpy()) < 1e-4


def test_mlp_and_lstm_train_mode_vs_oracle(cuda):
  import nnlib as nn
  rng = np.random.RandomState(11)
  model = {}
  dims = [20, 32, 12]
  run = nn.mlp(dims, [nn.relu, nn.softmax], phase_train=True, scope='tm', model=model)
  P = {}
  for i in range(2):
    P['tm_w_%d' % i] = (rng.randn(dims[i], dims[i + 1]) * 0.3).astype(np.float32)
    P['tm_b_%d' % i] = (rng.randn(dims[i + 1]) * 0.1).astype(np.float32)
    for n in ('w', 'b'):
      model['tm_%s_%d' % (n, i)].copy_(torch.from_numpy(P['tm_%s_%d' % (n, i)]))
      model['tm_%s_%d' % (n, i)].requires_grad_(True)
  x = rng.randn(5, 20).astype(np.float32)
  Pt = {k: _t64(v) for k, v in P.items()}
  xr = _t64(x)
  hr = ort.mlp(xr, Pt, 'tm', [torch.relu, lambda v: torch.softmax(v, dim=1)])
  cot = [rng.randn(*h.shape) for h in hr]
  sum((h * torch.tensor(c)).sum() for h, c in zip(hr, cot)).backward()
  xd = torch.tensor(x, device=cuda, requires_grad=True)
  hd = run(xd)
  sum((h * torch.tensor(c, dtype=torch.float32, device=cuda)).sum() for h, c in zip(hd, cot)).backward()
  for a, b in zip(hd, hr):
    assert _rel(a.detach().cpu().numpy(), b.detach().numpy()) < 1e-5
  assert _rel(xd.grad.cpu().numpy(), xr.grad.numpy()) < 1e-4
  for k in P:
    assert _rel(model[k].grad.cpu().numpy(), Pt[k].grad.numpy()) < 1e-4, k
  # eval-mode closure on the same weights (the fused dense kernel) agrees with the differentiable form
  run_e = nn.mlp(dims, [nn.relu, nn.softmax], phase_train=False, scope='tm2', model={},
                 init_weights=[{'w': P['tm_w_%d' % i], 'b': P['tm_b_%d' % i]} for i in range(2)])
  assert _rel(run_e(torch.tensor(x, device=cuda))[-1].cpu().numpy(), hr[-1].detach().numpy()) < 1e-5
  # dropout: identity at eval, kept values scaled by 1 / keep in training (nnlib.py:405-409)
  v = torch.ones(64, 64, device=cuda)
  assert nn.dropout(v, 0.5, False) is v
  d = nn.dropout(v, 0.5, True)
  assert set(np.unique(d.cpu().numpy()).tolist()) == {0.0, 2.0} and 0.3 < float((d > 0).float().mean()) < 0.7

  # ---- lstm (nnlib.py:637-649): state = [c | h], two chained cells, gradients to inputs and every parameter
  lm = {}
  cell = nn.lstm(6, 16, wd=None, scope='ctrl_lstm', model=lm)
  L = {}
  for k in list(lm):
    L[k] = (rng.randn(*lm[k].shape) * 0.4).astype(np.float32)
    lm[k].copy_(torch.from_numpy(L[k]))
    lm[k].requires_grad_(True)
  inp = rng.randn(3, 2, 6).astype(np.float32)
  st0 = (rng.randn(3, 32) * 0.5).astype(np.float32)
  Lt = {k: _t64(v) for k, v in L.items()}
  ir, sr = _t64(inp), _t64(st0)
  s = sr
  for t in range(2):
    s = ort.lstm(ir[:, t], s, Lt, 16)
  cs = rng.randn(3, 32)
  (s * torch.tensor(cs)).sum().backward()
  idv, sd = torch.tensor(inp, device=cuda, requires_grad=True), torch.tensor(st0, device=cuda, requires_grad=True)
  s2 = sd
  for t in range(2):
    s2, gi, gf, go = cell(idv[:, t], s2)
  (s2 * torch.tensor(cs, dtype=torch.float32, device=cuda)).sum().backward()
  assert _rel(s2.detach().cpu().numpy(), s.detach().numpy()) < 1e-5
  assert tuple(gi.shape) == (3, 16) and float(gi.min()) > 0 and float(go.max()) < 1
  assert _rel(idv.grad.cpu().numpy(), ir.grad.numpy()) < 1e-4 and _rel(sd.grad.cpu().numpy(), sr.grad.numpy()) < 1e-4
  for k in L:
    assert _rel(lm[k].grad.cpu().numpy(), Lt[k].grad.numpy()) < 1e-4, k
  # the no-grad closure (fused dense kernel) gives the same state
  with torch.no_grad():
    s3 = torch.tensor(st0, device=cuda)
    for t in range(2):
      s3 = cell(torch.tensor(inp[:, t], device=cuda), s3)[0]
  assert _rel(s3.cpu().numpy(), s.detach().numpy()) < 1e-5
