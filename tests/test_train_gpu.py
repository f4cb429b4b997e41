"""The training step (full_model.py:913-1057, phase_train = True, use_knob = False) on the HIP
kernels against torch autograd through the float64 CPU oracle (oracle/ra_oracle_torch.py): loss
pieces, the gradient of every parameter, three Adam steps and the BatchNorm EMA shadows."""
import numpy as np
import pytest
import torch

import ra_oracle as ora
import ra_oracle_torch as ort

pytestmark = pytest.mark.gpu


def _case(H=64, W=64, T=3, B=2, seed=3, **over):
  opt = ora.make_opt('cvppp', H, W, T, base_learn_rate=1e-3, learn_rate_decay=0.96, steps_per_learn_rate_decay=5000,
                     **over)
  P = ora.random_params(opt, seed)
  rng = np.random.RandomState(seed + 1)
  x = rng.rand(B, H, W, 3).astype(np.float32)
  y_gt, s_gt = np.zeros((B, T, H, W), np.float32), np.zeros((B, T), np.float32)
  for b in range(B):
    y_gt[b, 0, 6:30, 8:34] = 1
    y_gt[b, 1, 34:58, 30 + 2 * b:60] = 1
    s_gt[b, :2] = 1
  return opt, P, x, y_gt, s_gt


def _oracle_grads(opt, P, x, y_gt, s_gt):
  keys = [k for k in P if not (k.endswith('_ema_mean') or k.endswith('_ema_var'))]
  stats = {}
  fwd, Pt = ort.forward(opt, P, x, requires_grad=keys, phase_train=True, bn_stats=stats)
  head = ort.loss_head(opt, fwd, y_gt, s_gt)
  total = head['loss'] + ort.weight_decay_term(opt, {k: Pt[k] for k in keys})
  total.backward()
  return head, {k: Pt[k].grad.numpy() if Pt[k].grad is not None else np.zeros(P[k].shape) for k in keys}, stats


def _rel(a, b):
  return float(np.abs(a - b).max() / max(1e-7, np.abs(b).max()))


def test_conv_layer_forward_backward(cuda):
  """ConvBNActPool (conv + BN batch statistics + ReLU + pool, cnn and dcnn forms) against torch
  autograd on the same math."""
  import torch.nn.functional as F
  import ra_train
  rng = np.random.RandomState(0)
  for (cin, cout, pool, tr, stride, B, H, W) in [(4, 8, 1, False, 1, 2, 16, 24), (8, 16, 2, False, 1, 2, 16, 16),
                                                  (16, 32, 2, False, 1, 1, 8, 8), (32, 16, 1, True, 2, 2, 6, 6),
                                                  (8, 8, 1, True, 1, 2, 12, 12), (8, 1, 1, True, 1, 1, 16, 16),
                                                  (64, 64, 2, False, 1, 1, 8, 8)]:
    x = rng.randn(B, H, W, cin).astype(np.float32)
    w = (rng.randn(3, 3, cout, cin) if tr else rng.randn(3, 3, cin, cout)).astype(np.float32) * 0.2
    b, gam, bet = rng.randn(cout).astype(np.float32) * 0.1, rng.uniform(0.5, 1.5, cout).astype(np.float32), \
        rng.randn(cout).astype(np.float32) * 0.1
    t = lambda a, dt, dev: torch.tensor(a, dtype=dt, device=dev, requires_grad=True)
    # reference (float64, CPU)
    xr, wr, br, gr, ber = [t(a, torch.float64, 'cpu') for a in (x, w, b, gam, bet)]
    xi = xr.permute(0, 3, 1, 2)
    if tr:
      wt = wr.permute(3, 2, 0, 1)
      u = F.conv_transpose2d(xi, wt, stride=1, padding=1) if stride == 1 else \
          F.conv_transpose2d(xi, wt, stride=2, padding=0)[:, :, :2 * H, :2 * W]
    else:
      u = F.conv2d(xi, wr.permute(3, 2, 0, 1), padding=1)
    u = u.permute(0, 2, 3, 1) + br
    mean = u.mean(dim=(0, 1, 2))
    var = ((u - mean) ** 2).mean(dim=(0, 1, 2))
    v = torch.relu((u - mean) * torch.rsqrt(var + 1e-3) * gr + ber)
    yr = F.max_pool2d(v.permute(0, 3, 1, 2), pool, pool).permute(0, 2, 3, 1) if pool == 2 else v
    dy = rng.randn(*yr.shape).astype(np.float32)
    (yr * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    # product
    xd, wd, bd, gd, bed = [t(a, torch.float32, cuda) for a in (x, w, b, gam, bet)]
    meta = dict(transposed=tr, stride=stride, pool=pool, relu=True, chan_map=None)
    yd, md, vd = ra_train.ConvBNActPool.apply(xd, wd, bd, gd, bed, meta)
    (yd * torch.tensor(dy, device=cuda)).sum().backward()
    tag = (cin, cout, pool, tr, stride)
    assert _rel(yd.detach().cpu().numpy(), yr.detach().numpy()) < 1e-4, tag
    assert _rel(md.cpu().numpy(), mean.detach().numpy()) < 1e-4 and _rel(vd.cpu().numpy(), var.detach().numpy()) < 1e-4, tag
    for name, a, r in (('dx', xd, xr), ('dw', wd, wr), ('db', bd, br), ('dgamma', gd, gr), ('dbeta', bed, ber)):
      assert _rel(a.grad.cpu().numpy(), r.grad.numpy()) < 2e-3, (tag, name)


def test_pair_iou_gradient(cuda):
  import ra_train
  rng = np.random.RandomState(1)
  a = rng.rand(2, 3, 16, 20).astype(np.float32)
  b = (rng.rand(2, 4, 16, 20) > 0.6).astype(np.float32)
  g = rng.randn(2, 3, 4).astype(np.float32)
  ar = torch.tensor(a, dtype=torch.float64, requires_grad=True)
  iou = ort.iou_pairwise(ar, torch.tensor(b, dtype=torch.float64))
  (iou * torch.tensor(g, dtype=torch.float64)).sum().backward()
  ad = torch.tensor(a, device=cuda, requires_grad=True)
  out = ra_train.PairIoU.apply(ad, torch.tensor(b, device=cuda))
  (out * torch.tensor(g, device=cuda)).sum().backward()
  assert _rel(out.detach().cpu().numpy(), iou.detach().numpy()) < 1e-5
  assert _rel(ad.grad.cpu().numpy(), ar.grad.numpy()) < 1e-4


def test_loss_and_every_gradient_vs_oracle_autograd(cuda):
  import full_model
  import ra_train
  opt, P, x, y_gt, s_gt = _case()
  head, gref, stats = _oracle_grads(opt, P, x, y_gt, s_gt)
  m = full_model.get_model(opt).load_weights(P)
  ts = ra_train.TrainStep(m)
  ts.bucket.zero_grad()
  loss, pieces, st = ts.forward_loss(x, y_gt, s_gt)
  loss.backward()
  for k in ('loss', 'iou_soft', 'iou_soft_box', 'conf_loss'):
    assert abs(float(pieces[k]) - float(head[k])) < 2e-4 * max(1.0, abs(float(head[k]))), k
  assert (pieces['match'].cpu().numpy() == head['match'].numpy()).all()
  for key, (mean, var) in stats.items():
    assert _rel(st[key][0].cpu().numpy(), mean.numpy()) < 1e-3 and _rel(st[key][1].cpu().numpy(), var.numpy()) < 1e-3, key
  wd = float(opt['weight_decay'])
  worst = {}
  for k, g in gref.items():
    got = ts.bucket.grad_of[k].cpu().numpy()
    if ra_train.is_decayed(k):  # the product adds wd * w inside the optimizer kernel
      got = got + wd * P[k]
    scale = max(np.abs(g).max(), 1e-4)
    worst[k] = float(np.abs(got - g).max() / scale)
  bad = {k: v for k, v in worst.items() if v > 2e-2}
  assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]


def test_three_adam_steps_match_oracle(cuda):
  """model.run(['loss', 'train_step'], feed): weights after three steps against the oracle's own
  loop (autograd + TF-style Adam in float64), and the BN EMA shadows."""
  import full_model
  opt, P, x, y_gt, s_gt = _case(T=2)
  m = full_model.get_model(opt).load_weights(P)
  feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True}
  Pr = {k: v.astype(np.float64) for k, v in P.items()}
  mom = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in Pr.items()}
  for t in range(1, 4):
    loss_d, _ = m.run(['loss', 'train_step'], feed)
    head, gref, stats = _oracle_grads(opt, {k: v.astype(np.float32) for k, v in Pr.items()}, x, y_gt, s_gt)
    assert abs(float(loss_d) - float(head['loss'])) < 5e-3 * max(1.0, abs(float(head['loss']))), t
    lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
    for k, g in gref.items():
      g = np.clip(g, -1, 1)
      m1, v1 = mom[k]
      m1 = 0.9 * m1 + 0.1 * g
      v1 = 0.999 * v1 + 0.001 * g * g
      mom[k] = (m1, v1)
      Pr[k] = Pr[k] - lr_t * m1 / (np.sqrt(v1) + 1e-7)
    for key, (mean, var) in stats.items():
      Pr[key + '_ema_mean'] = 0.9 * Pr[key + '_ema_mean'] + 0.1 * mean.numpy()
      Pr[key + '_ema_var'] = 0.9 * Pr[key + '_ema_var'] + 0.1 * var.numpy()
  got = m.state_dict_numpy()
  worst = max(float(np.abs(got[k] - Pr[k]).max()) for k in Pr)
  # Adam normalises each step to ~lr: three steps move a weight by <= 3e-3, sign flips of tiny
  # gradients can cost a full step on a few elements; the bulk must agree far better
  frac_off = np.mean([np.mean(np.abs(got[k] - Pr[k]) > 2e-4) for k in Pr])
  assert worst < 7e-3 and frac_off < 0.02, (worst, frac_off)
  assert float(m['global_step']) == 3.0
