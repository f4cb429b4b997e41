"""The training step (full_model.py:913-1057, phase_train = True, use_knob = False) on the HIP
kernels against torch autograd through the float64 CPU oracle (oracle/ra_oracle_torch.py): loss
pieces, the gradient of every parameter, three Adam steps and the BatchNorm EMA shadows."""
import os

import numpy as np
import pytest
import torch

import ra_oracle as ora
import ra_oracle_torch as ort
import ra_train

pytestmark = pytest.mark.gpu


def _case(H=64, W=64, T=3, B=2, seed=3, wmul=1.0, **over):
  opt = ora.make_opt('cvppp', H, W, T, base_learn_rate=1e-3, learn_rate_decay=0.96, steps_per_learn_rate_decay=5000,
                     **over)
  P = ora.random_params(opt, seed)
  if wmul != 1.0:  # tame the layer gains: the comparison is float32 against float64 through ~40 BN layers
    for k in P:
      if ra_is_w(k):
        P[k] = (P[k] * wmul).astype(np.float32)
  rng = np.random.RandomState(seed + 1)
  x = rng.rand(B, H, W, 3).astype(np.float32)
  y_gt, s_gt = np.zeros((B, T, H, W), np.float32), np.zeros((B, T), np.float32)
  for b in range(B):
    y_gt[b, 0, H // 10:H // 2 - 2, W // 8:W // 2 + 2] = 1
    y_gt[b, 1, H // 2 + 4:H - 10, W // 2 - 2 + 2 * b:W - 4] = 1  # a different area: no tie in the greedy match
    s_gt[b, :2] = 1
  return opt, P, x, y_gt, s_gt


def ra_is_w(k):
  tail = k.split('_')
  return 'w' in tail or (len(tail[-1]) == 3 and tail[-1][0] == 'w' and tail[-1][1] in 'xh')


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
      got, ref = a.grad.cpu().numpy(), r.grad.numpy()
      if name == 'db':  # a bias in front of BatchNorm has zero gradient: both are round-off
        assert np.abs(got - ref).max() < 1e-3 * max(1.0, np.abs(dy).sum() ** 0.5), (tag, name)
      else:
        assert _rel(got, ref) < 2e-3, (tag, name)


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


def _pre_bn_bias(k):
  """conv biases feed straight into BatchNorm: their true gradient is exactly zero."""
  return '_cnn_b_' in k or '_dcnn_b_' in k


def _compare_grads(gref, got_of, P, wd):
  """Float32 kernels against float64 autograd through ~40 BatchNorm / ReLU / max-pool layers: kinks
  flip on round-off, so the bars are a per-tensor max-abs error relative to the tensor's own
  gradient scale and the cosine of the whole gradient vector (tools/train_debug3.py: a plain
  torch-float32 graph deviates from the oracle by the same few per cent)."""
  import ra_train
  dots = np.zeros(3)
  worst = {}
  gscale = max(np.abs(g).max() for g in gref.values())
  for k, g in gref.items():
    got = got_of(k)
    if ra_train.is_decayed(k):  # the product adds wd * w inside the optimizer kernel
      got = got + wd * P[k]
    if _pre_bn_bias(k):
      assert np.abs(got).max() < 2e-3 * gscale, k
      continue
    dots += [float((got * g).sum()), float((got * got).sum()), float((g * g).sum())]
    worst[k] = float(np.abs(got - g).max() / max(np.abs(g).max(), 1e-3 * gscale))
  cos = dots[0] / np.sqrt(dots[1] * dots[2])
  bad = {k: v for k, v in worst.items() if v > 5e-2}
  assert cos > 0.9995 and not bad, (cos, sorted(bad.items(), key=lambda kv: -kv[1])[:8])
  return cos, max(worst.values())


def test_loss_and_every_gradient_vs_oracle_autograd(cuda):
  import full_model
  import ra_train
  opt, P, x, y_gt, s_gt = _case(wmul=0.6)
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
  cos, worst = _compare_grads(gref, lambda k: ts.bucket.grad_of[k].cpu().numpy(), P, float(opt['weight_decay']))
  print('gradient cosine %.6f, worst per-tensor relative error %.3f' % (cos, worst))


def test_train_step_update_and_three_steps(cuda):
  """model.run(['loss', 'train_step'], feed) (full_model_train.py:107): the first optimizer step
  against the oracle's own update (autograd + TF-style Adam with +-1 clip in float64) and the BN EMA
  shadows; then two more steps: the schedule advances and the loss goes down."""
  import full_model
  opt, P, x, y_gt, s_gt = _case(T=2, wmul=0.6)
  m = full_model.get_model(opt).load_weights(P)
  feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': False}
  head, gref, stats = _oracle_grads(opt, P, x, y_gt, s_gt)
  loss1, _ = m.run(['loss', 'train_step'], feed)
  assert abs(float(loss1) - float(head['loss'])) < 2e-4 * max(1.0, abs(float(head['loss'])))
  got = m.state_dict_numpy()
  lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
  gscale = max(np.abs(g).max() for g in gref.values())
  off = tot = 0
  for k, g in gref.items():
    if _pre_bn_bias(k):
      continue
    gc = np.clip(g, -1, 1)
    ref = P[k].astype(np.float64) - lr_t * (0.1 * gc) / (np.sqrt(0.001 * gc * gc) + 1e-7)
    sure = np.abs(g) > 1e-3 * gscale  # Adam's first step is lr * sign(g): undecided where g ~ 0
    off += int((np.abs(got[k] - ref)[sure] > 1e-4).sum())
    tot += int(sure.sum())
    assert np.abs(got[k] - P[k]).max() <= 1.05e-3  # nothing moves by more than the learning rate
  assert tot > 1000 and off < 0.01 * tot, (off, tot)
  for key, (mean, var) in stats.items():  # shadow = 0.9 shadow + 0.1 batch statistic (nnlib.py:103-110)
    assert _rel(got[key + '_ema_mean'], 0.9 * P[key + '_ema_mean'] + 0.1 * mean.numpy()) < 1e-3, key
    assert _rel(got[key + '_ema_var'], 0.9 * P[key + '_ema_var'] + 0.1 * var.numpy()) < 1e-3, key
  losses = [float(loss1)] + [float(m.run(['loss', 'train_step'], feed)[0]) for _ in range(2)]
  assert float(m['global_step']) == 3.0 and all(np.isfinite(losses)) and losses[-1] < losses[0], losses
  # the decode path sees the trained weights (the optimizer kernel invalidates the packed copies)
  y = m.run('y_out', {'x': x, 'phase_train': False}, as_numpy=True)
  assert np.isfinite(y).all() and y.shape == (x.shape[0], 2, 64, 64)


KNOB_OPT = dict(use_knob=True, knob_base=1.0, knob_decay=0.9, steps_per_knob_decay=300, knob_box_offset=300,
                knob_segm_offset=500, knob_use_timescale=True, gt_box_ctr_noise=0.05, gt_box_pad_noise=0.1,
                gt_segm_noise=0.3)


@pytest.mark.parametrize('step,fixed,ioub', [(0, False, False), (700, False, False), (0, True, False), (0, False, True)],
                         ids=['knobs_on', 'knobs_decayed', 'fixed_order', 'use_iou_box'])
def test_knob_mixing_vs_oracle(cuda, step, fixed, ioub):
  """use_knob = True (run_cvppp.sh; full_model.py:559-625,744-773,826-841): noisy GT boxes, greedy
  per-timestep match, box / segmentation knobs — same draws into product and oracle: loss pieces
  and the whole gradient.  Step 700: both knob probabilities have decayed below 1, so some draws
  keep the prediction."""
  import full_model
  import ra_train
  opt, P, x, y_gt, s_gt = _case(T=3, wmul=0.6, fixed_order=fixed, use_iou_box=ioub, **KNOB_OPT)  # use_iou_box: Cityscapes' knob
  B, T, H, W = 2, 3, 64, 64
  rng = np.random.RandomState(5)
  knobs = {'pad': rng.uniform(0.1, 0.3, (B, T, 1)), 'shift': rng.uniform(-0.05, 0.05, (B, T, 2)),
           'u_box': rng.rand(B, T, 1), 'u_segm': rng.rand(B, T, 1), 'segm_noise': 0.3 * rng.rand(T, B, H, W)}
  keys = [k for k in P if not (k.endswith('_ema_mean') or k.endswith('_ema_var'))]
  fwd, Pt = ort.forward(opt, P, x, requires_grad=keys, phase_train=True, knobs=knobs, y_gt=y_gt, global_step=step)
  head = ort.loss_head(opt, fwd, y_gt, s_gt)
  (head['loss'] + ort.weight_decay_term(opt, {k: Pt[k] for k in keys})).backward()
  gref = {k: Pt[k].grad.numpy() if Pt[k].grad is not None else np.zeros(P[k].shape) for k in keys}
  m = full_model.get_model(opt).load_weights(P)
  ts = ra_train.TrainStep(m)
  ts.bucket.global_step = step
  kd = {k: torch.tensor(v, dtype=torch.float32, device=cuda) for k, v in knobs.items()}
  ts.bucket.zero_grad()
  loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt, knobs=kd)
  loss.backward()
  for k in ('loss', 'iou_soft', 'iou_soft_box', 'conf_loss'):
    assert abs(float(pieces[k]) - float(head[k])) < 3e-4 * max(1.0, abs(float(head[k]))), k
  assert (pieces['match'].cpu().numpy() == head['match'].numpy()).all()
  _compare_grads(gref, lambda k: ts.bucket.grad_of[k].cpu().numpy(), P, float(opt['weight_decay']))
  # the knob changes the graph: without it the loss is a different number
  loss0, _, _ = ts.forward_loss(x, y_gt, s_gt, knobs={k: (torch.ones_like(v) * 2 if k.startswith('u_') else v) for k, v in kd.items()})
  if step == 0:
    assert abs(float(loss0) - float(loss)) > 1e-3


@pytest.mark.parametrize('case', ['plain', 'knob_decayed', 'disable_overwrite'])
def test_canvas_gradient_vs_oracle(cuda, case):
  """stop_canvas_grad = False (full_model.py:843-848; every run script sets the flag, so this is the branch off the
  beaten path): canvas = max(y_c, canvas) stays differentiable — its gradient reaches every earlier mask through the
  next controller CNN's first layer (a data gradient in the packed input's canvas channel), the next extract (d X =
  gamma F_y dP F_x^T on the dense-bank operator) and disable_overwrite's (1 - canvas) factor.  Loss pieces and the
  whole gradient against float64 autograd; and the gradient differs from the stopped one (the branch is live)."""
  import full_model
  over = dict(stop_canvas_grad=False)
  if case == 'knob_decayed':
    over.update(KNOB_OPT)
  if case == 'disable_overwrite':
    over.update(disable_overwrite=True)
  # two timesteps: with three, this random network turns float32 round-off into 11 % in single BatchNorm gradients even
  # in a plain-torch float32 evaluation of the ORACLE's graph (the canvas gradient adds a path through every later step)
  # and a weight draw with no ReLU / max-pool kink within float32 round-off of flipping: with seed 3 a ONE-ulp change of the
  # output layer's batch mean at timestep 0 (a different summation order) moved single BatchNorm gradients of timestep 1 by
  # 29 % — in either direction of the change the float64 oracle cannot say which side float32 should land on
  # (tools/step_sequence.py era probe: seeds 5 and 9 agree to 0.4 % / 1.4 % with every summation order tried)
  opt, P, x, y_gt, s_gt = _case(T=2, wmul=0.6, seed=5, **over)
  B, T, H, W = 2, 2, 64, 64
  rng = np.random.RandomState(5)
  knobs, step = None, 0
  if case == 'knob_decayed':  # step 5000: knob probabilities 0.2, most draws keep the prediction (a live canvas gradient)
    step = 5000
    knobs = {'pad': rng.uniform(0.1, 0.3, (B, T, 1)), 'shift': rng.uniform(-0.05, 0.05, (B, T, 2)),
             'u_box': rng.rand(B, T, 1), 'u_segm': rng.rand(B, T, 1), 'segm_noise': 0.3 * rng.rand(T, B, H, W)}
  keys = [k for k in P if not (k.endswith('_ema_mean') or k.endswith('_ema_var'))]

  def oracle(o):
    fwd, Pt = ort.forward(o, P, x, requires_grad=keys, phase_train=True, knobs=knobs, y_gt=y_gt, global_step=step)
    head = ort.loss_head(o, fwd, y_gt, s_gt)
    (head['loss'] + ort.weight_decay_term(o, {k: Pt[k] for k in keys})).backward()
    return head, {k: Pt[k].grad.numpy() if Pt[k].grad is not None else np.zeros(P[k].shape) for k in keys}
  head, gref = oracle(opt)
  _, gstop = oracle(dict(opt, stop_canvas_grad=True))
  assert np.abs(gref['ctrl_cnn_w_0'] - gstop['ctrl_cnn_w_0']).max() > 1e-3 * np.abs(gstop['ctrl_cnn_w_0']).max()
  m = full_model.get_model(opt).load_weights(P)
  ts = ra_train.TrainStep(m)
  assert not ts._batched_ok([])   # the canvas gradient couples the timesteps: the per-timestep graph
  ts.bucket.global_step = step
  kd = None if knobs is None else {k: torch.tensor(v, dtype=torch.float32, device=cuda) for k, v in knobs.items()}
  ts.bucket.zero_grad()
  loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt, knobs=kd)
  loss.backward()
  for k in ('loss', 'iou_soft', 'iou_soft_box', 'conf_loss'):
    assert abs(float(pieces[k]) - float(head[k])) < 3e-4 * max(1.0, abs(float(head[k]))), k
  assert (pieces['match'].cpu().numpy() == head['match'].numpy()).all()
  got_of = lambda k: ts.bucket.grad_of[k].cpu().numpy()
  _compare_grads(gref, got_of, P, float(opt['weight_decay']))
  wd = float(opt['weight_decay'])
  assert _grad_cosine(gref, got_of, P, wd) > _grad_cosine(gstop, got_of, P, wd)  # closer to ITS oracle than to the stopped one
  feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': False}
  if kd is not None:
    feed['knobs'] = kd
  m2 = full_model.get_model(opt).load_weights(P)
  losses = [float(m2.run(['loss', 'train_step'], feed)[0]) for _ in range(3)]   # eager, captured, replayed
  assert all(np.isfinite(losses)), losses


CVPPP_FLAGS = ['--ctrl_add_inp', '--ctrl_add_canvas', '--attn_add_inp', '--attn_add_canvas', '--fixed_gamma',
               '--stop_canvas_grad', '--use_knob', '--knob_use_timescale',                       # run_cvppp.sh:44-72
               '--ctrl_cnn_filter_size', '3,3,3,3,3,3,3,3', '--ctrl_cnn_depth', '8,8,16,16,32,32,64,64',
               '--ctrl_cnn_pool', '1,2,1,2,1,2,2,2', '--attn_cnn_filter_size', '3,3,3,3,3,3',
               '--attn_cnn_depth', '8,8,16,16,32,32', '--attn_cnn_pool', '1,2,1,2,1,2',
               '--attn_dcnn_filter_size', '3,3,3,3,3,3,3', '--attn_dcnn_depth', '32,32,16,16,8,8,1',
               '--attn_dcnn_pool', '2,1,2,1,2,1,1']


def test_train_cli_then_eval_cli(cuda, tmp_path, capsys):
  """full_model_train.py WITHOUT --init_only: a few optimizer steps on synthetic CVPPP-shaped batches
  with the run script's flags, then full_model_eval.py restores what the trainer wrote."""
  import full_model_eval
  import full_model_train
  res = str(tmp_path / 'results')
  full_model_train.main(['--results', res, '--model_id', 't0', '--inp_height', '64', '--inp_width', '64',
                         '--timespan', '4', '--batch_size', '2', '--num_steps', '4', '--steps_per_log', '1'] + CVPPP_FLAGS)
  out = capsys.readouterr().out
  losses = [float(l.split('loss')[1].split()[0]) for l in out.splitlines() if l.startswith('step ')]
  assert len(losses) == 4 and all(np.isfinite(losses))
  w0 = dict(np.load(str(tmp_path / 'results' / 't0' / 'weights.npz')))
  assert all(np.isfinite(v).all() for v in w0.values())
  assert float(np.abs(w0['ctrl_cnn_0_0_ema_var']).sum()) > 0   # the EMA shadows moved off their zero init
  full_model_eval.main(['--model_id', 't0', '--results', res, '--num_synthetic', '2', '--batch_size', '2'])
  pred = np.load(str(tmp_path / 'results' / 't0' / 'output_valid' / 'pred_rank0.npz'))
  assert pred['y_out'].shape == (2, 4, 64, 64) and np.isfinite(pred['y_out']).all()


@pytest.mark.parametrize('over', [dict(box_loss_fn='mse'), dict(box_loss_fn='huber', segm_loss_fn='wt_cov')],
                         ids=['box_mse', 'box_huber_segm_wt_cov'])
def test_loss_variants_vs_oracle(cuda, over):
  """The other loss branches that execute in the reference: matched regression of the attention
  parameters (f_match_loss with f_squared_err / f_huber) and weighted coverage for the masks."""
  import full_model
  import ra_train
  opt, P, x, y_gt, s_gt = _case(wmul=0.6, **over)
  head, gref, _ = _oracle_grads(opt, P, x, y_gt, s_gt)
  m = full_model.get_model(opt).load_weights(P)
  ts = ra_train.TrainStep(m)
  ts.bucket.zero_grad()
  loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt)
  loss.backward()
  for k in ('loss', 'box_loss', 'segm_loss', 'conf_loss'):
    assert abs(float(pieces[k]) - float(head[k])) < 3e-4 * max(1.0, abs(float(head[k]))), k
  _compare_grads(gref, lambda k: ts.bucket.grad_of[k].cpu().numpy(), P, float(opt['weight_decay']))


@pytest.mark.parametrize('over', [dict(), dict(box_loss_fn='mse', fixed_order=True)], ids=['iou_matched', 'mse_fixed_order'])
def test_box_model_training_vs_oracle(cuda, over):
  """box_model's training graph (box_model.py:403-652): loss pieces and every gradient against the
  differentiable oracle with the same canvas-noise draws, then train_step through model.run."""
  import box_model
  import ra_train
  opt, _, x, y_gt, s_gt = _case(wmul=1.0, **over)
  P = ora.random_params(opt, 7, box_model=True)
  for k in P:
    if ra_is_w(k):
      P[k] = (P[k] * 0.6).astype(np.float32)
  noise = np.random.RandomState(9).uniform(0, 0.3, (3, 2, 64, 64)).astype(np.float32)
  keys = [k for k in P if not (k.endswith('_ema_mean') or k.endswith('_ema_var'))]
  head, Pt = ort.box_forward_loss(opt, P, x, y_gt, s_gt, noise, requires_grad=keys)
  (head['loss'] + ort.weight_decay_term(opt, {k: Pt[k] for k in keys})).backward()
  gref = {k: Pt[k].grad.numpy() if Pt[k].grad is not None else np.zeros(P[k].shape) for k in keys}
  m = box_model.get_model(opt).load_weights(P)
  ts = ra_train.BoxTrainStep(m)
  ts.bucket.zero_grad()
  loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt, knobs={'noise': noise})
  loss.backward()
  for k in ('loss', 'box_loss', 'conf_loss', 'iou_soft_box'):
    assert abs(float(pieces[k]) - float(head[k])) < 3e-4 * max(1.0, abs(float(head[k]))), k
  assert (pieces['match_box'].cpu().numpy() == head['match_box'].numpy()).all()
  _compare_grads(gref, lambda k: ts.bucket.grad_of[k].cpu().numpy(), P, float(opt['weight_decay']))
  m2 = box_model.get_model(opt).load_weights(P)
  feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'noise': noise, 'phase_train': True, 'aug': False}
  l = [float(m2.run(['loss', 'train_step'], feed)[0]) for _ in range(3)]
  assert abs(l[0] - float(head['loss'])) < 3e-4 * max(1.0, abs(float(head['loss']))) and l[2] < l[0] and float(m2['global_step']) == 3.0


def test_box_model_training_with_d_in_y_in_vs_oracle(cuda):
  """Stage 1 of the KITTI recipe (run_kitti.sh:45-59: box_model_train.py --add_d_out --add_y_out): box_model's training
  graph on concat(x, canvas, d_in, y_in) (box_model.py:404-410; 13 input channels through the first filter's channel map):
  loss pieces, matching and every gradient against the differentiable oracle with the same canvas noise, then the CLI."""
  import box_model
  import box_model_train
  import ra_train
  import tempfile
  H, W, T, B = 64, 96, 2, 2
  full = ora.make_opt('kitti', H, W, T, base_learn_rate=1e-3, learn_rate_decay=0.96, steps_per_learn_rate_decay=5000)
  opt = {k: full[k] for k in box_model_train.BOX_KEYS if k in full}
  opt.update(attn_box_padding_ratio=0.2, weight_decay=5e-5, use_bn=True, box_loss_fn='iou')
  P = ora.random_params(opt, 17, box_model=True)
  for k in P:
    if ra_is_w(k):
      P[k] = (P[k] * 0.5).astype(np.float32)
  assert P['ctrl_cnn_w_0'].shape[2] == 13
  rng = np.random.RandomState(18)
  x = rng.rand(B, H, W, 3).astype(np.float32)
  d_in = np.eye(8, dtype=np.float32)[rng.randint(0, 8, (B, H, W))]
  y_in = ora.softmax(rng.randn(B, H, W, 1)).astype(np.float32)
  y_gt, s_gt = np.zeros((B, T, H, W), np.float32), np.ones((B, T), np.float32)
  y_gt[:, 0, 6:30, 8:40] = 1
  y_gt[:, 1, 36:56, 50:90] = 1
  noise = rng.uniform(0, 0.3, (T, B, H, W)).astype(np.float32)
  keys = [k for k in P if not (k.endswith('_ema_mean') or k.endswith('_ema_var'))]
  head, Pt = ort.box_forward_loss(opt, P, x, y_gt, s_gt, noise, requires_grad=keys, d_in=d_in, y_in=y_in)
  (head['loss'] + ort.weight_decay_term(opt, {k: Pt[k] for k in keys})).backward()
  gref = {k: Pt[k].grad.numpy() if Pt[k].grad is not None else np.zeros(P[k].shape) for k in keys}
  assert np.abs(gref['ctrl_cnn_w_0'][:, :, 4:, :]).max() > 0  # the extra channels do reach the loss
  m = box_model.get_model(opt).load_weights(P)
  ts = ra_train.BoxTrainStep(m)
  ts.bucket.zero_grad()
  loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt, knobs={'noise': noise}, d_in=d_in, y_in=y_in)
  loss.backward()
  for k in ('loss', 'box_loss', 'conf_loss', 'iou_soft_box'):
    assert abs(float(pieces[k]) - float(head[k])) < 3e-4 * max(1.0, abs(float(head[k]))), k
  assert (pieces['match_box'].cpu().numpy() == head['match_box'].numpy()).all()
  _compare_grads(gref, lambda k: ts.bucket.grad_of[k].cpu().numpy(), P, float(opt['weight_decay']))
  with pytest.raises(Exception):
    ts.forward_loss(x, y_gt, s_gt, knobs={'noise': noise})  # d_in / y_in are part of this architecture's input
  m2 = box_model.get_model(opt).load_weights(P)
  feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'd_in': d_in, 'y_in': y_in, 'noise': noise, 'phase_train': True, 'aug': False}
  l = [float(m2.run(['loss', 'train_step'], feed)[0]) for _ in range(3)]
  assert abs(l[0] - float(head['loss'])) < 3e-4 * max(1.0, abs(float(head['loss']))) and l[2] < l[0]
  with tempfile.TemporaryDirectory() as res:  # the CLI with the run script's flags, synthetic d_in / y_in
    box_model_train.main(['--results', res, '--model_id', 'bk', '--num_steps', '2', '--batch_size', '2', '--inp_height', '64',
                          '--inp_width', '96', '--timespan', '2', '--add_d_out', '--add_y_out',
                          '--ctrl_cnn_filter_size', '3,3,3,3,3,3,3,3', '--ctrl_cnn_depth', '16,16,32,32,64,64,64,64',
                          '--ctrl_cnn_pool', '2,2,1,2,1,2,1,2'])
    w = dict(np.load(os.path.join(res, 'bk', 'weights.npz')))
    assert w['ctrl_cnn_w_0'].shape == (3, 3, 13, 16) and int(w['optim/global_step']) == 2


def test_two_stage_cli_box_then_full(cuda, tmp_path, capsys):
  """The reference's two-stage recipe (run_cvppp.sh:16-72): box_model_train.py pre-trains the
  controller, full_model_train.py --pretrain_ctrl_net starts from its weights."""
  import box_model_train
  import full_model_train
  res = str(tmp_path / 'results')
  size = ['--inp_height', '64', '--inp_width', '64', '--timespan', '3', '--batch_size', '2', '--steps_per_log', '1']
  ctrl = [f for f in CVPPP_FLAGS if 'attn' not in f]
  ctrl = [a for i, a in enumerate(CVPPP_FLAGS) if not (a.startswith('--attn') or (i and CVPPP_FLAGS[i - 1].startswith('--attn_')))]
  ctrl = [a for a in ctrl if a not in ('--use_knob', '--knob_use_timescale', '--stop_canvas_grad', '--fixed_gamma',
                                       '--ctrl_add_inp', '--ctrl_add_canvas')]
  box_model_train.main(['--results', res, '--model_id', 'b0', '--num_steps', '3'] + size + ctrl)
  out = capsys.readouterr().out
  assert out.count('step ') == 3 and 'weights ->' in out
  wb = dict(np.load(str(tmp_path / 'results' / 'b0' / 'weights.npz')))
  full_model_train.main(['--results', res, '--model_id', 'f0', '--num_steps', '2', '--pretrain_ctrl_net',
                         str(tmp_path / 'results' / 'b0' / 'weights.npz')] + size + CVPPP_FLAGS)
  out = capsys.readouterr().out
  assert out.count('step ') == 2
  wf = dict(np.load(str(tmp_path / 'results' / 'f0' / 'weights.npz')))
  # the LSTM came from the box model: two Adam steps move a weight by at most 2e-3
  assert np.abs(wf['ctrl_lstm_w_hi'] - wb['ctrl_lstm_w_hi']).max() < 2.5e-3
  assert np.abs(wb['ctrl_lstm_w_hi']).max() > 0.01


@pytest.mark.parametrize('knob', [False, True], ids=['plain', 'knob'])
def test_kitti_arch_training_vs_oracle(cuda, knob):
  """The KITTI-style training graph: 13 packed input channels (x, canvas, d_in, y_in) through channel
  maps, skip connections into the transposed-conv decoder (concat(prev, skip), nnlib.py:365),
  dynamic_var, free gammas, a 96-channel layer (backward-weight in two cout slices)."""
  import full_model
  import ra_train
  H, W, T, B = 64, 96, 2, 2
  over = dict(KNOB_OPT) if knob else {}
  opt = ora.make_opt('kitti', H, W, T, base_learn_rate=1e-3, learn_rate_decay=0.96, steps_per_learn_rate_decay=5000, **over)
  P = ora.random_params(opt, 13)
  for k in P:
    if ra_is_w(k):  # the deeper / wider KITTI stacks amplify float32 round-off more: tamer gains
      P[k] = (P[k] * 0.5).astype(np.float32)
  rng = np.random.RandomState(14)
  x = rng.rand(B, H, W, 3).astype(np.float32)
  d_in = np.eye(8, dtype=np.float32)[rng.randint(0, 8, (B, H, W))]
  y_in = ora.softmax(rng.randn(B, H, W, 1)).astype(np.float32)
  y_gt, s_gt = np.zeros((B, T, H, W), np.float32), np.ones((B, T), np.float32)
  y_gt[:, 0, 6:30, 8:40] = 1
  y_gt[:, 1, 36:56, 50:90] = 1
  knobs = None
  if knob:
    knobs = {'pad': rng.uniform(0.1, 0.3, (B, T, 1)), 'shift': rng.uniform(-0.05, 0.05, (B, T, 2)),
             'u_box': rng.rand(B, T, 1), 'u_segm': rng.rand(B, T, 1), 'segm_noise': 0.3 * rng.rand(T, B, H, W)}
  keys = [k for k in P if not (k.endswith('_ema_mean') or k.endswith('_ema_var'))]
  fwd, Pt = ort.forward(opt, P, x, d_in=d_in, y_in=y_in, requires_grad=keys, phase_train=True, knobs=knobs, y_gt=y_gt)
  head = ort.loss_head(opt, fwd, y_gt, s_gt)
  (head['loss'] + ort.weight_decay_term(opt, {k: Pt[k] for k in keys})).backward()
  gref = {k: Pt[k].grad.numpy() if Pt[k].grad is not None else np.zeros(P[k].shape) for k in keys}
  m = full_model.get_model(opt).load_weights(P)
  ts = ra_train.TrainStep(m)
  ts.bucket.zero_grad()
  kd = None if knobs is None else {k: torch.tensor(v, dtype=torch.float32, device=cuda) for k, v in knobs.items()}
  loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt, knobs=kd, d_in=d_in, y_in=y_in)
  loss.backward()
  for k in ('loss', 'iou_soft', 'iou_soft_box', 'conf_loss'):
    assert abs(float(pieces[k]) - float(head[k])) < 3e-4 * max(1.0, abs(float(head[k]))), k
  _compare_grads(gref, lambda k: ts.bucket.grad_of[k].cpu().numpy(), P, float(opt['weight_decay']))
  loss2, _ = m.run(['loss', 'train_step'], {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'd_in': d_in, 'y_in': y_in,
                                            'phase_train': True, 'knobs': kd, 'aug': False})
  assert abs(float(loss2) - float(head['loss'])) < 3e-4 * max(1.0, abs(float(head['loss'])))


def test_graphed_step_equals_eager_step(cuda):
  """TrainStep.use_graph: from the second step of a shape on, zero_grad + forward + backward + EMA are
  replayed from one HIP graph (inputs, random draws and the knob probabilities in static buffers).
  Five steps with knobs on a decaying schedule: losses, weights and BN shadows equal the eager run's."""
  import full_model
  import ra_train
  opt, P, x, y_gt, s_gt = _case(T=3, wmul=0.6, **dict(KNOB_OPT, steps_per_knob_decay=2, knob_box_offset=1, knob_segm_offset=1))
  rng = np.random.RandomState(11)
  B, T, H, W = x.shape[0], 3, 64, 64
  draws = [{'pad': rng.uniform(0.1, 0.3, (B, T, 1)), 'shift': rng.uniform(-0.05, 0.05, (B, T, 2)),
            'u_box': rng.rand(B, T, 1), 'u_segm': rng.rand(B, T, 1), 'segm_noise': 0.3 * rng.rand(T, B, H, W)}
           for _ in range(5)]
  xs = [x, x[::-1].copy(), x, x * 0.5, x]
  res = {}
  for graphed in (False, True):
    m = full_model.get_model(opt).load_weights(P)
    losses = []
    for k, xk in zip(draws, xs):
      if getattr(m, 'trainer', None) is not None:
        m.trainer.use_graph = graphed
      else:
        ra_train.TrainStep.use_graph = graphed
      loss, _ = m.run(['loss', 'train_step'], {'x': xk, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'knobs': k, 'aug': False})
      losses.append(float(loss))
    res[graphed] = (losses, m.state_dict_numpy())
    assert (graphed and any('graph' in v for v in m.trainer._graphs.values())) or \
        (not graphed and not any('graph' in v for v in m.trainer._graphs.values()))
  ra_train.TrainStep.use_graph = True
  assert np.allclose(res[True][0], res[False][0], rtol=1e-5, atol=1e-6), (res[True][0], res[False][0])
  for k, v in res[False][1].items():
    assert np.allclose(res[True][1][k], v, rtol=1e-4, atol=1e-6), k


def test_graphed_steps_survive_alternating_batch_shapes(cuda):
  """the step-persistent scratch (the [T, ...] slabs, the fused controller's buffers) is shared by all batch
  shapes and its addresses are inside a captured step.  Shapes A, A (captured), B, B (scratch moves), A, A again: the
  graphed run must equal the eager run step for step — a stale graph of shape A replayed after B's reallocation would
  read and write freed memory."""
  import full_model
  import ra_train
  opt, P, x, y_gt, s_gt = _case(T=2, B=3, wmul=0.6)
  sel = [slice(0, 3), slice(0, 3), slice(0, 1), slice(0, 1), slice(0, 3), slice(0, 3), slice(0, 1)]
  res = {}
  for graphed in (False, True):
    m = full_model.get_model(opt).load_weights(P)
    losses = []
    for k, sl in enumerate(sel):
      if getattr(m, 'trainer', None) is not None:
        m.trainer.use_graph = graphed
      else:
        ra_train.TrainStep.use_graph = graphed
      xk = (x * (1.0 - 0.05 * k)).astype(np.float32)
      loss, _ = m.run(['loss', 'train_step'], {'x': xk[sl], 'y_gt': y_gt[sl], 's_gt': s_gt[sl], 'phase_train': True, 'aug': False})
      losses.append(float(loss))
    res[graphed] = (losses, m.state_dict_numpy())
  ra_train.TrainStep.use_graph = True
  assert np.allclose(res[True][0], res[False][0], rtol=1e-5, atol=1e-6), (res[True][0], res[False][0])
  for k, v in res[False][1].items():
    assert np.allclose(res[True][1][k], v, rtol=1e-4, atol=1e-6), k


def test_fused_pointwise_kernels_vs_torch_autograd(cuda):
  """GaussFilter and LSTMCell (one kernel forward, one backward) against the same formulas under
  torch autograd in float64."""
  import math
  import ra_train
  g = torch.Generator().manual_seed(4)
  B, L, F = 3, 96, 16
  ctr = (torch.rand(B, generator=g) * L).double().requires_grad_(True)
  size = (10 + 40 * torch.rand(B, generator=g)).double().requires_grad_(True)
  lgv = (torch.randn(B, generator=g) * 0.5).double().requires_grad_(True)
  j = torch.arange(F, dtype=torch.float64)
  mu = ctr[:, None] + ((size[:, None] + 1.0) / F) * (j[None, :] - (F - 1) / 2.0)
  dd = torch.arange(L, dtype=torch.float64)[None, :, None] - mu[:, None, :]
  var = torch.exp(lgv)[:, None, None]
  ref = torch.exp(-0.5 * dd * dd / var) / (torch.sqrt(var) * math.sqrt(2 * math.pi))
  wgt = torch.randn(B, L, F, generator=g).double()
  (ref * wgt).sum().backward()
  c32, s32, v32 = [t.detach().float().to(cuda).requires_grad_(True) for t in (ctr, size, lgv)]
  out = ra_train.gaussian_filter(c32, s32, v32, L, F)
  (out * wgt.float().to(cuda)).sum().backward()
  assert np.abs(out.detach().cpu().numpy() - ref.detach().numpy()).max() < 1e-6
  for a, b in ((c32, ctr), (s32, size), (v32, lgv)):
    assert np.abs(a.grad.cpu().numpy() - b.grad.numpy()).max() < 2e-4 * max(1.0, np.abs(b.grad.numpy()).max())
  hid = 32
  pre = torch.randn(B, 4 * hid, generator=g).double().requires_grad_(True)
  c0 = torch.randn(B, hid, generator=g).double().requires_grad_(True)
  gi, gf, go, gu = [pre[:, k * hid:(k + 1) * hid] for k in range(4)]
  c1 = torch.sigmoid(gf) * c0 + torch.sigmoid(gi) * torch.tanh(gu)
  h1 = torch.sigmoid(go) * torch.tanh(c1)
  wh, wc = torch.randn(B, hid, generator=g).double(), torch.randn(B, hid, generator=g).double()
  ((h1 * wh).sum() + (c1 * wc).sum()).backward()
  p32, c32 = [t.detach().float().to(cuda).requires_grad_(True) for t in (pre, c0)]
  h, c = ra_train.LSTMCell.apply(p32, c32)
  ((h * wh.float().to(cuda)).sum() + (c * wc.float().to(cuda)).sum()).backward()
  assert np.abs(h.detach().cpu().numpy() - h1.detach().numpy()).max() < 1e-6
  assert np.abs(p32.grad.cpu().numpy() - pre.grad.numpy()).max() < 1e-5
  assert np.abs(c32.grad.cpu().numpy() - c0.grad.numpy()).max() < 1e-5
  p32.grad = None
  h, c = ra_train.LSTMCell.apply(p32, c32)  # c's gradient absent (the last glimpse of a timestep)
  (h * wh.float().to(cuda)).sum().backward()
  assert np.isfinite(p32.grad.cpu().numpy()).all()


@pytest.mark.parametrize('flags', [0, 1, 2, 5, 8, 9, 15])
def test_attention_head_and_knob_vs_torch_autograd(cuda, flags):
  """AttnHead (controller output -> window parameters and gammas), GaussFilterPair (both banks from the [B,2]
  tensors where they lie), KnobMix (the ground-truth knob) and LinearAcc (in-place parameter gradients): one launch
  each way, against the reference formulas (full_model.py:702-722,744-773, modellib.py:752-764,812-825) under torch
  autograd in float64.  flags: 1 squash, 2 fixed_var, 4 dynamic_var, 8 fixed_gamma."""
  import math
  import ra_train
  g = torch.Generator().manual_seed(10 + flags)
  B, T, H, W, Fh, Fw = 5, 4, 64, 96, 16, 16
  squash, fixed_var, dynamic_var, fixed_gamma = bool(flags & 1), bool(flags & 2), bool(flags & 4), bool(flags & 8)
  co = (torch.randn(B, 9, generator=g) * 0.7).double().requires_grad_(True)
  ctr_gt, size_gt = torch.rand(B, T, 2, generator=g).double() * 60, 10 + torch.rand(B, T, 2, generator=g).double() * 30
  match = torch.nn.functional.one_hot(torch.randint(0, T, (B,), generator=g), T).double()
  knob = (torch.rand(B, 3, 1, generator=g) > 0.5).double()
  dims, fdim = torch.tensor([H, W], dtype=torch.float64), torch.tensor([Fh, Fw], dtype=torch.float64)

  def bank(c, s, lv, L, F):
    j = torch.arange(F, dtype=torch.float64)
    mu = c[:, None] + ((s[:, None] + 1.0) / F) * (j[None, :] - (F - 1) / 2.0)
    dd = torch.arange(L, dtype=torch.float64)[None, :, None] - mu[:, None, :]
    var = torch.exp(lv)[:, None, None]
    return torch.exp(-0.5 * dd * dd / var) / (torch.sqrt(var) * math.sqrt(2 * math.pi))

  cn, ls = co[:, 0:2], co[:, 2:4]
  if squash:
    cn, ls = torch.tanh(cn), -torch.nn.functional.softplus(ls)
  ctr, size = (cn + 1.0) * dims / 2.0, torch.exp(ls) * dims
  lv = torch.zeros_like(ctr) if fixed_var else torch.log(size) - torch.log(fdim)
  if dynamic_var:
    lv = co[:, 4:6]
  ag = torch.ones(B, dtype=torch.float64) if fixed_gamma else torch.exp(co[:, 6])
  ylg = torch.full((B,), 2.0, dtype=torch.float64) if fixed_gamma else co[:, 8]
  bg = torch.exp(co[:, 7])
  kb = knob[:, 1]
  ctr2 = kb * (match[:, :, None] * ctr_gt).sum(dim=1) + (1 - kb) * ctr
  size2 = kb * (match[:, :, None] * size_gt).sum(dim=1) + (1 - kb) * size
  fy, fx = bank(ctr2[:, 0], size2[:, 0], lv[:, 0], H, Fh), bank(ctr2[:, 1], size2[:, 1], lv[:, 1], W, Fw)
  fy0 = bank(ctr[:, 0], size[:, 0], lv[:, 0], H, Fh)
  wy, wx, w0 = [torch.randn(*s, generator=g).double() for s in ((B, H, Fh), (B, W, Fw), (B, H, Fh))]
  ws = torch.randn(B, 7, generator=g).double()
  ref = (fy * wy).sum() + (fx * wx).sum() + (fy0 * w0).sum() + (ag * ws[:, 0]).sum() + (ylg * ws[:, 1]).sum() + \
      (bg * ws[:, 2]).sum() + (cn * ws[:, 3:5]).sum() + (ls * ws[:, 5:7]).sum()
  ref.backward()

  f = lambda t: t.detach().float().to(cuda)
  co32 = f(co).requires_grad_(True)
  cn_, ls_, ctr_, size_, lv_, ag_, bg_, ylg_, arec = ra_train.AttnHead.apply(co32, H, W, Fh, Fw, flags)
  c2, s2, arec2 = ra_train.KnobMix.apply(ctr_, size_, f(match), f(ctr_gt), f(size_gt), f(knob)[:, 1], arec)
  # the attention records the two kernels write = the record torch.cat builds from the same fields (one for all three gammas)
  with torch.no_grad():
    want = ra_train.attn_record(ctr_, size_, lv_, attn_gamma=ag_, box_gamma=bg_, y_lg_gamma=ylg_)
    assert torch.equal(arec, want) and not arec.requires_grad
    assert torch.equal(arec2, ra_train.attn_record(c2, s2, lv_, attn_gamma=ag_, box_gamma=bg_, y_lg_gamma=ylg_))
  fy_, fx_ = ra_train.gaussian_filters(c2, s2, lv_, H, W, Fh, Fw)
  fy0_, _ = ra_train.gaussian_filters(ctr_, size_, lv_, H, W, Fh, Fw)
  wsd = f(ws)
  out = (fy_ * f(wy)).sum() + (fx_ * f(wx)).sum() + (fy0_ * f(w0)).sum() + (ag_ * wsd[:, 0]).sum() + (ylg_ * wsd[:, 1]).sum() + \
      (bg_ * wsd[:, 2]).sum() + (cn_ * wsd[:, 3:5]).sum() + (ls_ * wsd[:, 5:7]).sum()
  out.backward()
  for a, b in ((ctr_, ctr), (size_, size), (lv_, lv), (ag_, ag), (bg_, bg), (ylg_, ylg), (c2, ctr2), (s2, size2), (fy_, fy), (fx_, fx)):
    assert np.abs(a.detach().cpu().numpy() - b.detach().numpy()).max() < 2e-5 * max(1.0, float(b.detach().abs().max()))
  gr, gd = co.grad.numpy(), co32.grad.cpu().numpy()
  assert np.abs(gd - gr).max() < 5e-4 * max(1.0, np.abs(gr).max()), np.abs(gd - gr).max()

  # LinearAcc: y = x W + b with dW, db added in place to caller buffers, dx returned
  x = torch.randn(B, 12, generator=g).double().requires_grad_(True)
  Wl, bl = torch.randn(12, 7, generator=g).double().requires_grad_(True), torch.randn(7, generator=g).double().requires_grad_(True)
  wo = torch.randn(B, 7, generator=g).double()
  ((x @ Wl + bl) * wo).sum().backward()
  x32 = f(x).requires_grad_(True)
  gw, gb = torch.ones(12, 7, device=cuda), torch.ones(7, device=cuda)   # accumulate on top of what is there
  y = ra_train.LinearAcc.apply(x32, f(Wl).requires_grad_(True), f(bl).requires_grad_(True), gw, gb)
  (y * f(wo)).sum().backward()
  assert np.abs(x32.grad.cpu().numpy() - x.grad.numpy()).max() < 1e-4
  assert np.abs(gw.cpu().numpy() - 1.0 - Wl.grad.numpy()).max() < 1e-4
  assert np.abs(gb.cpu().numpy() - 1.0 - bl.grad.numpy()).max() < 1e-4


def test_training_graph_applies_the_augmentation(cuda):
  """full_model.py:203-232: img.random_transformation(x, padding, phase_train, ..., y=y_gt) sits in front of the
  training graph.  With the step's draws given (crop offset, flips, transpose), the product's loss equals the
  oracle's loss on the oracle-transformed batch; without `aug` the trainer draws its own (rank-offset) offsets."""
  import full_model
  opt, P, x, y_gt, s_gt = _case(T=2, wmul=0.6, rnd_hflip=True, rnd_vflip=True, rnd_transpose=True)
  pad = int(opt['padding'])
  for draws in (dict(off_y=3, off_x=2 * pad - 1, flip_v=True, flip_h=False, transpose=True),
                dict(off_y=pad, off_x=pad, flip_v=False, flip_h=True, transpose=False)):
    xt = ora.random_transformation(x, pad, **draws)
    yt = np.stack([ora.random_transformation(y_gt[:, t][..., None], pad, **draws)[..., 0] for t in range(y_gt.shape[1])], axis=1)
    fwd, _ = ort.forward(opt, P, xt, phase_train=True, bn_stats={})
    head = ort.loss_head(opt, fwd, yt, s_gt)
    m = full_model.get_model(opt).load_weights(P)
    loss, _ = m.run(['loss', 'train_step'], {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': draws})
    assert abs(float(loss) - float(head['loss'])) < 3e-4 * max(1.0, abs(float(head['loss']))), draws
    assert m.trainer.last_aug['off_y'] == draws['off_y'] and m.trainer.last_aug['transpose'] == draws['transpose']
  m = full_model.get_model(opt).load_weights(P)
  seen = set()
  for _ in range(4):
    loss, _ = m.run(['loss', 'train_step'], {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True})
    a = m.trainer.last_aug
    assert 0 <= a['off_y'] < 2 * pad and 0 <= a['off_x'] < 2 * pad and np.isfinite(float(loss))
    seen.add((a['off_y'], a['off_x'], a['flip_v'], a['flip_h'], a['transpose']))
  assert len(seen) > 1  # the draws change from step to step


def test_resume_equals_uninterrupted(cuda, tmp_path):
  """utils/saver.py:24-31 saves every variable (weights, Adam slots, global_step), experiment.py:26-37 restores
  them: 2 steps + checkpoint + 2 steps in a fresh model equal 4 steps in one run, bit for bit (same inputs)."""
  import full_model
  import full_model_train
  opt, P, x, y_gt, s_gt = _case(T=2, wmul=0.6)
  opt['steps_per_learn_rate_decay'] = 3  # the staircase moves inside the run: a restart from step 0 would show
  feeds = [{'x': x * (1.0 - 0.1 * k), 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': False} for k in range(4)]
  a = full_model.get_model(opt).load_weights(P)
  la = [float(a.run(['loss', 'train_step'], f)[0]) for f in feeds]
  b = full_model.get_model(opt).load_weights(P)
  lb = [float(b.run(['loss', 'train_step'], f)[0]) for f in feeds[:2]]
  path = str(tmp_path / 'ckpt.npz')
  full_model_train.save_checkpoint(path, b)
  c = full_model.get_model(opt)           # fresh random weights, fresh optimizer
  full_model_train.load_checkpoint(path, c)
  assert c.trainer.bucket.global_step == 2 and float(c['global_step']) == 2.0
  lc = [float(c.run(['loss', 'train_step'], f)[0]) for f in feeds[2:]]
  assert lb == la[:2] and lc == la[2:], (la, lb, lc)
  wa, wc = a.state_dict_numpy(), c.state_dict_numpy()
  assert all((wa[k] == wc[k]).all() for k in wa)
  sa, sc = a.trainer.state_dict(), c.trainer.state_dict()
  assert all((sa[k] == sc[k]).all() for k in sa)


def test_sync_bn_split_backward_equals_fused(cuda):
  """The two-launch-group form of the BatchNorm backward (reduce | all-reduce | dx with the global count) that
  data-parallel training on whole-batch statistics uses: for ONE rank holding the whole batch it must equal the
  fused form; and two half-batches with whole-batch moments + summed per-channel sums reproduce the whole batch."""
  import ctypes as C
  import ra_native as rn
  rng = np.random.RandomState(3)
  B, H, W, Cc = 4, 8, 12, 8
  u = torch.tensor(rng.randn(B, H, W, Cc).astype(np.float32), device=cuda)
  dy = torch.tensor(rng.randn(B, H // 2, W // 2, Cc).astype(np.float32), device=cuda)
  gamma = torch.tensor(rng.uniform(0.5, 1.5, Cc).astype(np.float32), device=cuda)
  beta = torch.tensor(rng.randn(Cc).astype(np.float32) * 0.1, device=cuda)
  lib, f = rn.lib(), lambda *s: torch.empty(s, dtype=torch.float32, device=cuda)
  ws = f(lib.ra_bn_workspace_floats(Cc))

  def moments(t):
    mean, var = f(Cc), f(Cc)
    rn.check(lib.ra_bn_moments_f32(rn.ptr(t), t.numel() // Cc, Cc, rn.ptr(ws), ws.numel(), rn.ptr(mean), rn.ptr(var), rn.stream_ptr()), 'm')
    return mean, var
  mean, var = moments(u)
  dg, db, du = f(Cc), f(Cc), torch.empty_like(u)
  rn.check(lib.ra_bn_act_pool_bwd_f32(rn.ptr(u), rn.ptr(dy), rn.ptr(mean), rn.ptr(var), rn.ptr(gamma), rn.ptr(beta), C.c_float(1e-3),
                                      1, 2, B, H, W, Cc, rn.ptr(ws), ws.numel(), rn.ptr(dg), rn.ptr(db), rn.ptr(du), rn.stream_ptr()), 'bwd')
  # the same through combine_moments + reduce / dx on two shards
  halves = [(u[:2].contiguous(), dy[:2].contiguous()), (u[2:].contiguous(), dy[2:].contiguous())]
  ms = [moments(h[0]) for h in halves]
  n, gm, gv = ra_train.combine_moments(torch.tensor([2.0 * H * W, 2.0 * H * W], device=cuda), torch.stack([m[0] for m in ms]),
                                       torch.stack([m[1] for m in ms]))
  assert float(n) == B * H * W and np.abs(gm.cpu().numpy() - mean.cpu().numpy()).max() < 1e-6 and \
      np.abs(gv.cpu().numpy() - var.cpu().numpy()).max() < 1e-6
  sums = []
  for uh, dyh in halves:
    dgl, dbl = f(Cc), f(Cc)
    rn.check(lib.ra_bn_act_pool_bwd_reduce_f32(rn.ptr(uh), rn.ptr(dyh), rn.ptr(gm), rn.ptr(gv), rn.ptr(gamma), rn.ptr(beta),
                                               C.c_float(1e-3), 1, 2, 2, H, W, Cc, rn.ptr(ws), ws.numel(), rn.ptr(dgl), rn.ptr(dbl),
                                               None, None, rn.stream_ptr()), 'reduce')
    sums.append((dgl, dbl))
  dgs, dbs = sums[0][0] + sums[1][0], sums[0][1] + sums[1][1]
  assert np.abs(dgs.cpu().numpy() - dg.cpu().numpy()).max() < 1e-4 and np.abs(dbs.cpu().numpy() - db.cpu().numpy()).max() < 1e-4
  outs = []
  for uh, dyh in halves:
    duh = torch.empty_like(uh)
    rn.check(lib.ra_bn_act_pool_bwd_dx_f32(rn.ptr(uh), rn.ptr(dyh), rn.ptr(gm), rn.ptr(gv), rn.ptr(gamma), rn.ptr(beta), rn.ptr(dgs),
                                           rn.ptr(dbs), C.c_double(float(B * H * W)), C.c_float(1e-3), 1, 2, 2, H, W, Cc, rn.ptr(duh),
                                           rn.stream_ptr()), 'dx')
    outs.append(duh)
  assert np.abs(torch.cat(outs).cpu().numpy() - du.cpu().numpy()).max() < 1e-5


def test_sync_bn_stacked_backward_two_shards_equal_whole_batch(cuda):
  """--sync_bn on the STACKED backward (ConvStackFn with sync_world > 1: per layer the T groups' 2 C sums cross the ranks
  in ONE all-reduce, 21 collectives per step instead of 21 T): the product's node run for two half-batch "ranks" — their
  all-reduce replaced by the sum of the two ranks' recorded sums — against the same node on the whole batch: the data
  gradient of the two shards side by side, and filter / bias / gamma / beta gradients added over the shards."""
  rng = np.random.RandomState(21)
  G, Bw, Hs, Ws, cin, cout, pool = 2, 4, 8, 12, 8, 8, 2
  f = lambda a: torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device=cuda)
  X = rng.randn(G, Bw, Hs, Ws, cin).astype(np.float32)
  U = rng.randn(G, Bw, Hs, Ws, cout).astype(np.float32)
  dY = rng.randn(G, Bw, Hs // pool, Ws // pool, cout).astype(np.float32)
  w = f(rng.randn(3, 3, cin, cout) * 0.2)
  b = f(np.zeros(cout))
  gam = [f(rng.uniform(0.5, 1.5, cout)) for _ in range(G)]
  bet = [f(rng.randn(cout) * 0.1) for _ in range(G)]
  means = [f(U[g].reshape(-1, cout).mean(0)) for g in range(G)]   # whole-batch moments of each timestep group
  vars_ = [f(U[g].reshape(-1, cout).var(0)) for g in range(G)]

  def run(sel, sync_world, hook):
    Bs = len(sel)
    Xs, Us, dYs = [f(a[:, sel].reshape((G * Bs,) + a.shape[2:])) for a in (X, U, dY)]
    Xs.requires_grad_(True)
    gw, gb = torch.zeros_like(w), torch.zeros(cout, device=cuda)
    gg, gbt = [torch.zeros(cout, device=cuda) for _ in range(G)], [torch.zeros(cout, device=cuda) for _ in range(G)]
    cols = [means, vars_, gam, bet, gg, gbt]
    tab = torch.tensor([t.data_ptr() for col in cols for t in col], dtype=torch.int64).to(cuda)
    info = dict(G=G, B=Bs, U=Us, Y=torch.zeros_like(dYs), transposed=False, stride=1, pool=pool, relu=True, chan_map=None, bf16=False,
                tabs=tab, per_group=list(zip(*cols)), gw=gw, gb=gb, cache={}, sync_world=sync_world)
    orig = ra_train.allreduce_sums
    ra_train.allreduce_sums = hook
    try:
      y = ra_train.ConvStackFn.apply(Xs, w, b, info)
      y.backward(dYs)
    finally:
      ra_train.allreduce_sums = orig
    torch.cuda.synchronize()
    return Xs.grad.view((G, Bs) + Xs.shape[1:]).cpu().numpy(), [t.cpu().numpy() for t in [gw, gb] + gg + gbt]

  def no_collective(t):
    raise AssertionError('the whole-batch node must not ask for a collective')
  dx_ref, p_ref = run([0, 1, 2, 3], 1, no_collective)
  shards, seen = ([0, 1], [2, 3]), []

  def record(t):
    seen.append(t.clone())
    return t
  for sel in shards:  # what each rank would send
    run(sel, 2, record)
  total = seen[0] + seen[1]
  assert total.shape == (G, 2 * cout)   # ONE collective per layer: all T groups' sums together

  def summed(t):
    t.copy_(total)
    return t
  outs = [run(sel, 2, summed) for sel in shards]
  dx = np.concatenate([o[0] for o in outs], axis=1)
  assert np.abs(dx - dx_ref).max() < 2e-5 * max(1.0, np.abs(dx_ref).max()), np.abs(dx - dx_ref).max()
  for a0, a1, r in zip(outs[0][1], outs[1][1], p_ref):
    assert np.abs(a0 + a1 - r).max() < 1e-4 * max(1.0, np.abs(r).max()), np.abs(a0 + a1 - r).max()


def test_two_trainers_do_not_share_step_caches(cuda):
  """The per-step caches (packed filters, padded biases, packed LSTM weights) belong to the TrainStep: two
  trainers whose steps interleave give the losses they give alone."""
  import full_model
  opt, P, x, y_gt, s_gt = _case(T=2, wmul=0.6)
  P2 = ora.random_params(opt, 9)
  for k in P2:
    if ra_is_w(k):
      P2[k] = (P2[k] * 0.6).astype(np.float32)
  feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': False}
  alone = []
  for PP in (P, P2):
    m = full_model.get_model(opt).load_weights(PP)
    alone.append([float(m.run(['loss', 'train_step'], feed)[0]) for _ in range(3)])
  ma, mb = full_model.get_model(opt).load_weights(P), full_model.get_model(opt).load_weights(P2)
  inter = [[], []]
  for _ in range(3):
    inter[0].append(float(ma.run(['loss', 'train_step'], feed)[0]))
    inter[1].append(float(mb.run(['loss', 'train_step'], feed)[0]))
  assert inter == alone and ma.trainer._pack is not mb.trainer._pack


@pytest.mark.parametrize('H,W,C,wide', [(64, 96, 4, False), (128, 128, 8, True), (40, 56, 4, False)])
def test_banded_resample_adjoints_vs_dense_autograd(cuda, H, W, C, wide):
  """AttnExtract / AttnPaste (box, read, write on the banded kernels, backward straight to the window parameters:
  csrc/ra_attn_train.hip) against the reference's formulation — dense [L,F] Gaussian banks, F_y^T X F_x /
  F_y P F_x^T as matrix products — under torch autograd in float64."""
  import math
  g = torch.Generator().manual_seed(H + C)
  B, Fh, Fw = 3, 16, 12
  r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
  ctr = torch.stack([r(B) * H * 0.6 + 0.2 * H, r(B) * W * 0.6 + 0.2 * W], dim=1)
  ctr[0, 0] = 0.03 * H  # a window partly outside the image
  size = torch.stack([r(B) * H * 0.5 + 8, r(B) * W * 0.5 + 8], dim=1)
  lgv = (torch.randn(B, 2, generator=g, dtype=torch.float64) * 0.4 + (1.5 if wide else 0.3))
  gam = r(B) + 0.5
  x = r(B, H, W, C)
  P = torch.randn(B, Fh, Fw, generator=g, dtype=torch.float64)
  wE, wB, wY = torch.randn(B, Fh, Fw, C, generator=g, dtype=torch.float64), torch.randn(B, H, W, generator=g, dtype=torch.float64), \
      torch.randn(B, H, W, generator=g, dtype=torch.float64)

  def banks(ctr, size, lgv):
    out = []
    for ax, (L, F) in enumerate(((H, Fh), (W, Fw))):
      j = torch.arange(F, dtype=torch.float64)
      mu = ctr[:, ax, None] + ((size[:, ax, None] + 1.0) / F) * (j[None, :] - (F - 1) / 2.0)
      dd = torch.arange(L, dtype=torch.float64)[None, :, None] - mu[:, None, :]
      var = torch.exp(lgv[:, ax])[:, None, None]
      out.append(torch.exp(-0.5 * dd * dd / var) / (torch.sqrt(var) * math.sqrt(2 * math.pi)))
    return out

  leaves = [t.clone().requires_grad_(True) for t in (ctr, size, lgv, gam, P)]
  c64, s64, v64, g64, P64 = leaves
  fy, fx = banks(c64, s64, v64)
  e_ref = g64[:, None, None, None] * torch.einsum('blj,blwc,bwi->bjic', fy, x, fx)
  b_ref = torch.sigmoid(g64[:, None, None] * torch.einsum('blj,bwi->blw', fy, fx) - 5.0)
  y_ref = torch.sigmoid(torch.exp(g64 - 1.0)[:, None, None] * torch.einsum('blj,bji,bwi->blw', fy, P64, fx) - 5.0)
  ((e_ref * wE).sum() + (b_ref * wB).sum() + (y_ref * wY).sum()).backward()

  l32 = [t.detach().float().to(cuda).requires_grad_(True) for t in (ctr, size, lgv, gam, P)]
  c32, s32, v32, g32, P32 = l32
  e = ra_train.AttnExtract.apply(x.float().to(cuda), c32, s32, v32, g32, Fh, Fw)
  bx = ra_train.AttnPaste.apply(None, c32, s32, v32, g32, H, W, Fh, Fw)
  y = ra_train.AttnPaste.apply(P32[..., None], c32, s32, v32, g32 - 1.0, H, W, Fh, Fw)
  ((e * wE.float().to(cuda)).sum() + (bx * wB.float().to(cuda)).sum() + (y * wY.float().to(cuda)).sum()).backward()
  for got, ref in ((e, e_ref), (bx, b_ref), (y, y_ref)):
    assert np.abs(got.detach().cpu().numpy() - ref.detach().numpy()).max() < 3e-5 * max(1.0, float(ref.abs().max()))
  for name, a, b in zip(('ctr', 'size', 'lg_var', 'gamma', 'patch'), l32, leaves):
    ga, gb = a.grad.cpu().numpy(), b.grad.numpy()
    assert np.abs(ga - gb).max() < 1e-4 * max(1.0, np.abs(gb).max()), (name, np.abs(ga - gb).max(), np.abs(gb).max())


@pytest.mark.parametrize('knob', [False, True], ids=['plain', 'knob'])
def test_fused_controller_equals_library_path(cuda, knob):
  """ControllerFn (one forward + one backward launch per timestep, parameter gradients as four GEMMs per step over the
  saved rows; csrc/ra_ctrl_train.hip) against the same graph on library GEMMs / element-wise ops under autograd: loss
  and every gradient.  (The library path itself is checked against the float64 oracle above.)"""
  import full_model
  # two timesteps and a weight draw away from ReLU / pool kinks (see test_canvas_gradient_vs_oracle): the two sides differ by
  # float32 summation order in the controller, and every further timestep multiplies what a flipped kink does to single tensors
  opt, P, x, y_gt, s_gt = _case(T=2, wmul=0.6, seed=5, **(KNOB_OPT if knob else {}))
  rng = np.random.RandomState(5)
  B, T, H, W = 2, 2, 64, 64
  kd = None
  if knob:
    kd = {k: torch.tensor(v, dtype=torch.float32, device=cuda) for k, v in
          {'pad': rng.uniform(0.1, 0.3, (B, T, 1)), 'shift': rng.uniform(-0.05, 0.05, (B, T, 2)), 'u_box': rng.rand(B, T, 1),
           'u_segm': rng.rand(B, T, 1), 'segm_noise': 0.3 * rng.rand(T, B, H, W)}.items()}
  res = {}
  for fused in (False, True):
    m = full_model.get_model(opt).load_weights(P)
    ts = ra_train.TrainStep(m)
    ts.fuse_controller = fused
    for rep in range(2):  # twice: the step buffers are reused, nothing of the first pass may leak into the second
      ts.bucket.zero_grad()
      loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt, knobs=kd)
      loss.backward()
    assert (getattr(ts, '_ctl', None) is not None) == fused
    res[fused] = (float(loss), {k: ts.bucket.grad_of[k].cpu().numpy().copy() for k in ts.bucket.names})
  assert abs(res[True][0] - res[False][0]) < 2e-4 * max(1.0, abs(res[False][0]))  # float32 through recurrent timesteps, different summation orders
  scale = max(np.abs(g).max() for g in res[False][1].values())
  dots = np.zeros(3)
  for k, g in res[False][1].items():  # both sides are float32 graphs through ~40 BN / ReLU / pool layers per timestep
    got = res[True][1][k]
    err = np.abs(got - g).max()
    assert err < 0.1 * max(np.abs(g).max(), 1e-2 * scale), (k, err, np.abs(g).max())  # ReLU / pool kinks downstream of a 1e-6 different window flip
    dots += [float((got * g).sum()), float((got * got).sum()), float((g * g).sum())]
  cos = dots[0] / np.sqrt(dots[1] * dots[2])
  assert cos > 0.99999, cos
  for k in ('ctrl_lstm_w_xi', 'ctrl_lstm_w_hu', 'ctrl_lstm_b_f', 'glimpse_mlp_w_0', 'glimpse_mlp_b_1', 'glimpse_mlp_w_1', 'ctrl_mlp_w_0',
            'ctrl_mlp_b_0', 'ctrl_cnn_w_7'):  # the controller's own parameters and what feeds it: tight
    g, got = res[False][1][k], res[True][1][k]
    assert np.abs(got - g).max() < 1e-3 * max(np.abs(g).max(), 1e-4 * scale), (k, np.abs(got - g).max(), np.abs(g).max())


@pytest.mark.parametrize('n_g,n_c', [(2, 1), (1, 1), (3, 2), (4, 3)])
def test_controller_fn_unit_vs_library(cuda, n_g, n_c):
  """ControllerFn alone: one timestep's controller on a random feature map, forward outputs and — for random upstream
  gradients of h and ctrl_out — d feat and every controller parameter's gradient, against the library-GEMM path; at the
  run scripts' depths (2 glimpse-MLP layers, 1 controller-MLP layer) and at others (full_model.py:350-352,382-384)."""
  import full_model
  opt, P, x, y_gt, s_gt = _case(T=2, wmul=1.0, num_glimpse_mlp_layers=n_g, num_ctrl_mlp_layers=n_c, ctrl_mlp_dim=48)
  rng = np.random.RandomState(2)
  for k in P:  # livelier controller weights than the 0.01 init: the softmax must not stay uniform
    if k.startswith(('ctrl_lstm_w', 'glimpse_mlp_w', 'ctrl_mlp_w')):
      P[k] = (rng.randn(*P[k].shape) * 0.15).astype(np.float32)
    if k.startswith(('ctrl_lstm_b', 'glimpse_mlp_b', 'ctrl_mlp_b')):
      P[k] = (rng.randn(*P[k].shape) * 0.1).astype(np.float32)
  B = 3
  res = {}
  for fused in (False, True):
    m = full_model.get_model(opt).load_weights(P)
    ts = ra_train.TrainStep(m)
    ts.fuse_controller = fused
    d = ts.d
    assert (d['n_gmlp'], d['n_cmlp']) == (n_g, n_c) and (ts._ctrl_buffers(B, 64) is not None) == fused
    feat = torch.tensor(np.random.RandomState(4).rand(B, d['G'], 64).astype(np.float32), device=cuda, requires_grad=True)
    wh = torch.tensor(np.random.RandomState(5).randn(B, d['hid']).astype(np.float32), device=cuda)
    wc = torch.tensor(np.random.RandomState(6).randn(B, 9).astype(np.float32), device=cuda)
    ts.bucket.zero_grad()
    ts._pack.clear()
    h, co = ts._controller(feat, 1)
    ((h * wh).sum() + (co * wc).sum()).backward()
    names = [k for k in ts.bucket.names if k.startswith(('ctrl_lstm', 'glimpse_mlp', 'ctrl_mlp'))]
    res[fused] = (h.detach().cpu().numpy(), co.detach().cpu().numpy(), feat.grad.cpu().numpy(),
                  {k: ts.bucket.grad_of[k].cpu().numpy().copy() for k in names})
  (h0, c0, f0, g0), (h1, c1, f1, g1) = res[False], res[True]
  assert np.abs(h1 - h0).max() < 2e-6 and np.abs(c1 - c0).max() < 2e-6 * max(1.0, np.abs(c0).max())
  assert np.abs(f1 - f0).max() < 1e-5 * max(1.0, np.abs(f0).max()), np.abs(f1 - f0).max()
  for k in g0:
    assert np.abs(g1[k] - g0[k]).max() < 2e-5 * max(1.0, np.abs(g0[k]).max()), (k, np.abs(g1[k] - g0[k]).max(), np.abs(g0[k]).max())


@pytest.mark.parametrize('dtype', ['float32', 'bf16'])
def test_cfg4_shapes_training_step_properties(cuda, dtype):
  """BASELINE.json configs[3] at its per-GPU shapes — CVPPP arch, 512x512, T = 16, B = 8, use_knob, the random crop,
  in float32 AND in the configuration's stated bf16 (model_opt['compute_dtype']) — too large for the float64 oracle, so
  properties: three optimisation steps (eager, captured, replayed) give finite losses, every parameter moves by at most
  the learning rate per step, the BN shadows move towards the batch statistics, the fused controller and the banded
  resample ran, the trainer is in the mode asked for, and a captured replay equals the eager step it copies."""
  import full_model
  import full_model_train as fmt
  opt = ora.make_opt('cvppp', 512, 512, 16, base_learn_rate=1e-3, learn_rate_decay=0.96, steps_per_learn_rate_decay=5000,
                     compute_dtype=dtype, **KNOB_OPT)
  res = {}
  for graphed in (True, False):
    torch.manual_seed(3)
    m = full_model.get_model(opt)
    w0 = m.state_dict_numpy()
    rng = np.random.RandomState(11)
    x, y_gt, s_gt = fmt.synthetic_batch(rng, 8, 512, 512, 16)
    gen = torch.Generator(device='cuda').manual_seed(5)
    feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'generator': gen, 'aug': dict(off_y=3, off_x=29)}
    losses = []
    for k in range(3):
      if getattr(m, 'trainer', None) is not None:
        m.trainer.use_graph = graphed
      else:
        ra_train.TrainStep.use_graph = graphed
      losses.append(float(m.run(['loss', 'train_step'], feed)[0]))
    ra_train.TrainStep.use_graph = True
    w3 = m.state_dict_numpy()
    res[graphed] = (losses, w3)
    assert all(np.isfinite(losses)), losses
    assert m.trainer._ctl is not None and m.trainer.bucket.global_step == 3
    assert m.trainer.bf16 == (dtype == 'bf16')
    assert all(t.dtype == torch.float32 for t in (m.trainer.bucket.param, m.trainer.bucket.m, m.trainer.bucket.v))  # master state
    moved = 0
    for k, v in w3.items():
      assert np.isfinite(v).all(), k
      if k.endswith('_ema_mean') or k.endswith('_ema_var'):
        continue
      assert np.abs(v - w0[k]).max() <= 3.2e-3, k  # three Adam steps of at most lr each (clip +-1, lr 1e-3)
      moved += int(np.abs(v - w0[k]).max() > 0)
    assert moved > 200
    assert np.abs(w3['ctrl_cnn_3_0_ema_var']).max() > 0 and np.abs(w3['attn_dcnn_2_5_ema_mean']).max() > 0
  assert np.allclose(res[True][0], res[False][0], rtol=2e-4, atol=1e-5), (res[True][0], res[False][0])


@pytest.mark.parametrize('dtype', ['float32', 'bf16', 'bf16s'])
def test_full_resolution_controller_cnn_stack_pin(cuda, dtype):
  """A pin at cfg4's RESOLUTION that no attention decision feeds, so the bar can be tight: the controller CNN of
  timestep 0 — eight stacked conv + BatchNorm(batch moments, nnlib.py:98) + ReLU + pool layers on concat(x, canvas = 0),
  512x512, B = 8, training mode — through the product's training forward against the float64 oracle's conv stack
  (oracle/ra_oracle_torch.cnn, in bf16 mode with the operand roundings of set_conv_operands): every layer's batch mean
  and variance within 1e-3 of the channel scale.  'bf16s': the stacked step's form, U / Y stored as bf16 between the layers,
  against the oracle with the same storage roundings."""
  import full_model
  H = W = 512
  B = 8
  store = dtype == 'bf16s'
  dtype = 'bf16' if store else dtype
  opt = ora.make_opt('cvppp', H, W, 1, base_learn_rate=1e-3, learn_rate_decay=0.96, steps_per_learn_rate_decay=5000,
                     compute_dtype=dtype)
  P = ora.random_params(opt, 5)
  for k in P:
    if ra_is_w(k):
      P[k] = (P[k] * 0.6).astype(np.float32)
  rng = np.random.RandomState(6)
  x = rng.rand(B, H, W, 3).astype(np.float32)
  y_gt, s_gt = np.zeros((B, 1, H, W), np.float32), np.ones((B, 1), np.float32)
  y_gt[:, 0, 100:300, 150:380] = 1
  stats = {}
  ort.set_conv_operands(('bf16s' if store else 'bf16') if dtype == 'bf16' else None)
  ort._BN.update(train=True, stats=stats)
  try:
    with torch.no_grad():
      Pt = {k: ort.t64(v) for k, v in P.items() if k.startswith('ctrl_cnn_')}
      inp = torch.cat([ort.t64(x), torch.zeros((B, H, W, 1), dtype=torch.float64)], dim=3)
      feat = ort.cnn(inp, Pt, 'ctrl_cnn', 8, opt['ctrl_cnn_pool'], 0, True)[-1].numpy()
  finally:
    ort._BN.update(train=False, stats=None)
    ort.set_conv_operands(None)
  m = full_model.get_model(opt).load_weights(P)
  ts = ra_train.TrainStep(m)
  assert ts.bf16 == (dtype == 'bf16')
  if store:  # the stacked step (it needs autograd on; nothing is differentiated here)
    _, _, st = ts.forward_loss(x, y_gt, s_gt)
    assert ts.bf16_store and ts._slabs['ctrl_cnn_0_u'].dtype == torch.bfloat16
  else:
    with torch.no_grad():
      _, _, st = ts.forward_loss(x, y_gt, s_gt)
    assert not ts.bf16_store
  worst = 0.0
  for i in range(8):
    key = 'ctrl_cnn_%d_0' % i
    em, ev = _rel(st[key][0].cpu().numpy(), stats[key][0].numpy()), _rel(st[key][1].cpu().numpy(), stats[key][1].numpy())
    worst = max(worst, em, ev)
    assert em < 1e-3 and ev < 1e-3, (dtype, key, em, ev)
  print('controller-CNN stack at 512x512, B=8, %s: worst batch-moment deviation from the float64 oracle %.2e' % (dtype, worst))
  assert feat.shape == (B, 16, 16, 64)


# ---- mixed precision: model_opt['compute_dtype'] = 'bf16' (conv forward / data gradient / filter gradient with
# bf16 operands on the bf16 MFMA, float32 accumulation; everything else float32)
def _grad_cosine(gref, got_of, P, wd):
  dots = np.zeros(3)
  for k, g in gref.items():
    if not _pre_bn_bias(k):
      got = got_of(k) + (wd * P[k] if ra_train.is_decayed(k) else 0.0)
      dots += [float((got * g).sum()), float((got * got).sum()), float((g * g).sum())]
  return dots[0] / np.sqrt(dots[1] * dots[2])


def test_bf16_conv_layer_vs_emulating_oracle(cuda):
  """One ConvBNActPool layer (cnn and dcnn forms) in bf16 mode against float64 autograd through the oracle's conv with
  bf16-rounded operands (ra_oracle_torch.set_conv_operands: forward conv(R(x), R(w)), data gradient convT(R(du), R(w)),
  filter / bias gradients corr(R(x), R(du)), sum R(du)).  Outputs and batch statistics: 1e-4 (max norm, the float32
  layer's bar); gradients: 2e-3 in relative L2 norm (the float32 layer's bar; L2 because a du that sits on a bf16
  rounding boundary rounds the other way in float32 than in float64 and moves single elements by 2^-8).  Against the
  UNROUNDED float64 layer the distance is bf16's own: 2e-2 for outputs, 1e-1 (L2) for gradients, where ReLU / max-pool
  decisions flip."""
  import torch.nn.functional as F
  rng = np.random.RandomState(1)
  for (cin, cout, pool, tr, stride, B, H, W) in [(4, 8, 1, False, 1, 2, 16, 24), (8, 16, 2, False, 1, 2, 16, 16),
                                                  (16, 32, 2, False, 1, 1, 8, 8), (32, 16, 1, True, 2, 2, 6, 6),
                                                  (8, 8, 1, True, 1, 2, 12, 12), (8, 1, 1, True, 1, 1, 16, 16),
                                                  (64, 64, 2, False, 1, 1, 8, 8)]:
    x = rng.randn(B, H, W, cin).astype(np.float32)
    w = (rng.randn(3, 3, cout, cin) if tr else rng.randn(3, 3, cin, cout)).astype(np.float32) * 0.2
    b, gam, bet = rng.randn(cout).astype(np.float32) * 0.1, rng.uniform(0.5, 1.5, cout).astype(np.float32), \
        rng.randn(cout).astype(np.float32) * 0.1
    t = lambda a, dt, dev: torch.tensor(a, dtype=dt, device=dev, requires_grad=True)
    dy = None
    refs = {}
    for kind in ('bf16', None):
      ort.set_conv_operands(kind)
      try:
        xr, wr, br, gr, ber = [t(a, torch.float64, 'cpu') for a in (x, w, b, gam, bet)]
        u = ort.deconv_same(xr, wr, br, stride) if tr else ort.conv_same(xr, wr, br)
        mean = u.mean(dim=(0, 1, 2))
        var = ((u - mean) ** 2).mean(dim=(0, 1, 2))
        v = torch.relu((u - mean) * torch.rsqrt(var + 1e-3) * gr + ber)
        yr = F.max_pool2d(v.permute(0, 3, 1, 2), pool, pool).permute(0, 2, 3, 1) if pool == 2 else v
        if dy is None:
          dy = rng.randn(*yr.shape).astype(np.float32)
        (yr * torch.tensor(dy, dtype=torch.float64)).sum().backward()
      finally:
        ort.set_conv_operands(None)
      refs[kind] = dict(y=yr.detach().numpy(), mean=mean.detach().numpy(), var=var.detach().numpy(), dx=xr.grad.numpy(),
                        dw=wr.grad.numpy(), dgamma=gr.grad.numpy(), dbeta=ber.grad.numpy())
    xd, wd, bd, gd, bed = [t(a, torch.float32, cuda) for a in (x, w, b, gam, bet)]
    meta = dict(transposed=tr, stride=stride, pool=pool, relu=True, chan_map=None, bf16=True)
    yd, md, vd = ra_train.ConvBNActPool.apply(xd, wd, bd, gd, bed, meta)
    (yd * torch.tensor(dy, device=cuda)).sum().backward()
    got = dict(y=yd.detach().cpu().numpy(), mean=md.cpu().numpy(), var=vd.cpu().numpy(), dx=xd.grad.cpu().numpy(),
               dw=wd.grad.cpu().numpy(), dgamma=gd.grad.cpu().numpy(), dbeta=bed.grad.cpu().numpy())
    tag = (cin, cout, pool, tr, stride)
    for k in got:
      l2 = lambda r: float(np.linalg.norm(got[k] - r) / np.linalg.norm(r))
      if k in ('y', 'mean', 'var'):
        assert _rel(got[k], refs['bf16'][k]) < 1e-4, (tag, k, _rel(got[k], refs['bf16'][k]))
        assert _rel(got[k], refs[None][k]) < 2e-2, (tag, k, _rel(got[k], refs[None][k]))
      else:
        assert l2(refs['bf16'][k]) < 2e-3, (tag, k, l2(refs['bf16'][k]))
        assert l2(refs[None][k]) < 1e-1, (tag, k, l2(refs[None][k]))
    assert np.abs(got['y'] - refs[None]['y']).max() > 1e-5 * np.abs(got['y']).max(), tag  # not the float32 kernels


def test_bf16_training_step_vs_emulating_oracle(cuda):
  """The whole training step with model_opt['compute_dtype'] = 'bf16'.  Its oracle is the float64 graph with the conv
  layers' three products on bf16-rounded operands (ra_oracle_torch.set_conv_operands).  What the comparison can show
  is bounded by the test network, not by the kernels: a randomly initialised recurrent hard-attention net amplifies
  a perturbation of the conv outputs by ~1e4 on its way through box -> crop -> next timestep (the float32 step
  agrees with float64 to 1e-3 on its BN statistics from 6e-8 round-off), and the emulation cannot be closer than
  the operands that round the other way in float32 than in float64 (one in ~1e4, each by 2^-8).  So:
    - the controller CNN of timestep 0 — eight stacked conv + BN + pool layers that no attention decision feeds —
      matches the emulating oracle's batch statistics to 1e-3 (5e-3 in its last two, few-pixel layers): the forward
      kernels do what the mode says;
    - on a two-timestep network every loss piece is within 3e-2, the matching is the same and the whole gradient's
      cosine is above 0.95 (0.98 measured; the layer test above pins the backward kernels to 2e-3) — and the step is
      closer to ITS oracle than to the unrounded float64 one (cosine 0.96), which is what tells an emulated rounding
      from an error.  tools/bf16_step_probe.py prints the same numbers for other depths, gains and seeds (T = 3 at
      this gain: 0.78 against 0.45 — and the float32 step itself is at 0.66 from the emulating oracle).
  Master weights, Adam state and the checkpoint stay float32, three optimizer steps bring the loss down, an unknown
  compute_dtype is an error."""
  import full_model
  opt, P, x, y_gt, s_gt = _case(wmul=0.6, T=2)
  head64, gref64, _ = _oracle_grads(opt, P, x, y_gt, s_gt)
  ort.set_conv_operands('bf16')
  try:
    head, gref, stats = _oracle_grads(opt, P, x, y_gt, s_gt)
  finally:
    ort.set_conv_operands(None)
  opt_b = dict(opt, compute_dtype='bf16')
  m = full_model.get_model(opt_b).load_weights(P)
  ts = ra_train.TrainStep(m)
  assert ts.bf16
  ts.bf16_storage = False  # the operand-only form (the storage form has its own oracle: test_bf16_storage_step_vs_emulating_oracle)
  ts.bucket.zero_grad()
  loss, pieces, st = ts.forward_loss(x, y_gt, s_gt)
  loss.backward()
  assert not ts.bf16_store
  for k in ('loss', 'iou_soft', 'iou_soft_box', 'conf_loss'):
    got = float(pieces[k].detach())
    assert abs(got - float(head[k])) < 3e-2 * max(1.0, abs(float(head[k]))), (k, got, float(head[k]))
  assert (pieces['match'].cpu().numpy() == head['match'].numpy()).all()
  n0 = 0
  for key, (mean, var) in stats.items():
    em, ev = _rel(st[key][0].cpu().numpy(), mean.numpy()), _rel(st[key][1].cpu().numpy(), var.numpy())
    if key.startswith('ctrl_cnn_') and key.endswith('_0'):
      bar = 1e-3 if int(key.split('_')[2]) < 6 else 5e-3  # the two 1/16-resolution layers average over few pixels
      assert em < bar and ev < bar, (key, em, ev)
      n0 += 1
    assert em < 0.2 and ev < 0.2, (key, em, ev)
  assert n0 == 8
  wd = float(opt['weight_decay'])
  got_of = lambda k: ts.bucket.grad_of[k].cpu().numpy()
  cos, cos64 = _grad_cosine(gref, got_of, P, wd), _grad_cosine(gref64, got_of, P, wd)
  print('bf16 step vs its oracle: gradient cosine %.4f, loss %.6f vs %.6f;  vs the unrounded float64 oracle: cosine %.4f, '
        'loss %.6f' % (cos, float(pieces['loss'].detach()), float(head['loss']), cos64, float(head64['loss'])))
  assert cos > 0.95 and cos > cos64, (cos, cos64)
  assert all(t.dtype == torch.float32 for t in (ts.bucket.param, ts.bucket.grad, ts.bucket.m, ts.bucket.v))
  feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': False}
  m2 = full_model.get_model(opt_b).load_weights(P)
  losses = [float(m2.run(['loss', 'train_step'], feed)[0]) for _ in range(3)]
  assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
  with pytest.raises(Exception):
    ra_train.TrainStep(full_model.get_model(dict(opt, compute_dtype='fp8')).load_weights(P))


def test_bf16_storage_step_vs_emulating_oracle(cuda):
  """The bf16 mode as the stacked step runs it (round 4): on top of the bf16 operands, U / Y / dY / dU of the conv layers
  are STORED as bf16 between their passes (csrc: ra_conv3x3_bf16_f32, ra_bn_act_pool_bf16_f32, the grouped BatchNorm
  backward and the filter gradient with storage flags; tests/test_bf16_storage_gpu.py pins each kernel bit for bit to
  its float32-storage form).  Its oracle is the float64 graph with the same roundings (ra_oracle_torch 'bf16s').  On the
  two-timestep random network: the timestep-0 controller CNN's statistics to 1e-3 (5e-3 in the two few-pixel layers),
  loss pieces to 3e-2, the same matching, and the gradient closer to ITS oracle than to the operand-only oracle and to
  the unrounded one (cosine 0.87 / 0.81 / 0.77 measured; the operand-only step reaches 0.98 against its own: twice the
  rounding points on a network that amplifies any of them ~1e4-fold, tools/bf16_step_probe.py).  The slabs are bf16, master
  weights and Adam state float32, and three optimizer steps run eager, captured and replayed."""
  import full_model
  opt, P, x, y_gt, s_gt = _case(wmul=0.6, T=2)
  orc = {}
  for kind in (None, 'bf16', 'bf16s'):
    ort.set_conv_operands(kind)
    try:
      orc[kind] = _oracle_grads(opt, P, x, y_gt, s_gt)
    finally:
      ort.set_conv_operands(None)
  head, gref, stats = orc['bf16s']
  opt_b = dict(opt, compute_dtype='bf16')
  m = full_model.get_model(opt_b).load_weights(P)
  ts = ra_train.TrainStep(m)
  ts.seq_ctrl_split = False
  ts.bucket.zero_grad()
  loss, pieces, st = ts.forward_loss(x, y_gt, s_gt)
  loss.backward()
  assert ts.bf16 and ts.bf16_store
  assert ts._slabs['ctrl_cnn_0_u'].dtype == torch.bfloat16 and ts._slabs['ctrl_cnn_3_y'].dtype == torch.bfloat16
  assert ts._slabs['ctrl_cnn_7_y'].dtype == torch.float32 and ts._slabs['attn_dcnn_6_u'].dtype == torch.float32  # float32 readers / 1 channel
  for k in ('loss', 'iou_soft', 'iou_soft_box', 'conf_loss'):
    got = float(pieces[k].detach())
    assert abs(got - float(head[k])) < 3e-2 * max(1.0, abs(float(head[k]))), (k, got, float(head[k]))
  assert (pieces['match'].cpu().numpy() == head['match'].numpy()).all()
  n0 = 0
  for key, (mean, var) in stats.items():
    if key.startswith('ctrl_cnn_') and key.endswith('_0'):
      em, ev = _rel(st[key][0].cpu().numpy(), mean.numpy()), _rel(st[key][1].cpu().numpy(), var.numpy())
      bar = 1e-3 if int(key.split('_')[2]) < 6 else 5e-3
      assert em < bar and ev < bar, (key, em, ev)
      n0 += 1
  assert n0 == 8
  wd = float(opt['weight_decay'])
  got_of = lambda k: ts.bucket.grad_of[k].cpu().numpy()
  cos = {kind: _grad_cosine(orc[kind][1], got_of, P, wd) for kind in orc}
  print('bf16 storage step: gradient cosine vs its oracle %.4f, vs the operand-only oracle %.4f, vs float64 %.4f' % (cos['bf16s'], cos['bf16'], cos[None]))
  assert cos['bf16s'] > 0.8 and cos['bf16s'] > cos['bf16'] and cos['bf16s'] > cos[None], cos
  assert all(t.dtype == torch.float32 for t in (ts.bucket.param, ts.bucket.grad, ts.bucket.m, ts.bucket.v))
  feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': False}
  m2 = full_model.get_model(opt_b).load_weights(P)
  losses = [float(m2.run(['loss', 'train_step'], feed)[0]) for _ in range(3)]
  assert all(np.isfinite(losses)) and losses[-1] < losses[0] and m2.trainer.bf16_store, losses


@pytest.mark.parametrize('B,H,W,Ci,Co,ups,bf', [(2, 32, 32, 4, 8, 0, 0), (1, 33, 35, 8, 16, 0, 0), (8, 48, 48, 16, 32, 0, 1),
                                               (2, 8, 12, 32, 16, 1, 0), (1, 128, 160, 8, 8, 0, 1), (3, 24, 24, 64, 64, 0, 0),
                                               (4, 256, 256, 4, 8, 0, 0)])
def test_conv_epilogue_moments(cuda, B, H, W, Ci, Co, ups, bf):
  """ra_conv3x3_moments_f32 + ra_bn_moments_from_partials_f32 (the batch moments of nnlib.py:98 out of the conv
  epilogue): the conv output equals the plain kernel's bit for bit, mean / var equal float64 moments of that output to
  1e-5 of the channel's scale — also in a channel whose mean is 300x its spread (sums about a per-wave pivot, Chan's
  combination: no E[x^2] - E[x]^2) — and the two-pass kernels' to 1e-5."""
  import ctypes as C
  import ra_native as rn
  import ra_ops as ops
  rng = np.random.RandomState(B + H + Ci + Co)
  Hs, Ws = (H // 2, W // 2) if ups else (H, W)
  x = torch.tensor(rng.randn(B, Hs, Ws, Ci).astype(np.float32), device=cuda)
  w = (rng.randn(3, 3, Ci, Co) / np.sqrt(9 * Ci)).astype(np.float32)
  b = (rng.randn(Co) * 0.1).astype(np.float32)
  b[0] = 300.0  # |mean| >> spread
  wp = torch.tensor(ops.pack_conv_weights(w), device=cuda)
  sc, sh = ops.fold_bn(b, Co, None)
  sc, sh = torch.tensor(sc, device=cuda), torch.tensor(sh, device=cuda)
  u0 = ops.conv3x3(x, wp, sc, sh, Co, relu=False, pool=1, upsample=bool(ups), bf16=bool(bf))
  u = torch.empty_like(u0)
  part = torch.empty(rn.lib().ra_conv3x3_moments_part_floats(Co), device=cuda)
  nparts = C.c_int(0)
  mean, var = torch.empty(Co, device=cuda), torch.empty(Co, device=cuda)
  ops.check(rn.lib().ra_conv3x3_moments_f32(ops.ptr(x), Ci, None, 0, B, Hs, Ws, ups, ops.ptr(wp), ops.ptr(sc), ops.ptr(sh), Co, 0, bf,
                                            ops.ptr(u), ops.ptr(part), part.numel(), C.byref(nparts), rn.stream_ptr()), 'moments conv')
  ops.check(rn.lib().ra_bn_moments_from_partials_f32(ops.ptr(part), nparts.value, Co, ops.ptr(mean), ops.ptr(var), rn.stream_ptr()),
            'moments finish')
  m2, v2 = torch.empty(Co, device=cuda), torch.empty(Co, device=cuda)
  ws = torch.empty(rn.lib().ra_bn_workspace_floats(Co), device=cuda)
  ops.check(rn.lib().ra_bn_moments_f32(ops.ptr(u0), B * H * W, Co, ops.ptr(ws), ws.numel(), ops.ptr(m2), ops.ptr(v2), rn.stream_ptr()),
            'two-pass moments')
  torch.cuda.synchronize()
  assert nparts.value > 0 and torch.equal(u, u0)
  ud = u0.double().reshape(-1, Co).cpu().numpy()
  rm, rv = ud.mean(axis=0), ud.var(axis=0)
  scale = np.sqrt(rv)
  assert (np.abs(mean.cpu().numpy() - rm) < 1e-5 * np.maximum(scale, np.abs(rm))).all(), np.abs(mean.cpu().numpy() - rm) / scale
  assert (np.abs(var.cpu().numpy() - rv) < 1e-5 * rv + 1e-9).all(), np.abs(var.cpu().numpy() - rv) / rv
  assert (np.abs(v2.cpu().numpy() - rv) < 1e-4 * rv + 1e-9).all()


@pytest.mark.parametrize('B,T,H,W', [(2, 3, 64, 64), (8, 16, 128, 96), (1, 21, 48, 160)])
def test_box_iou_against_rectangles(cuda, B, T, H, W):
  """ra_box_iou_rects_f32 (the per-timestep f_iou row of the training graph, full_model.py:744-758) against the pairwise
  statistics kernel on the filled rectangles: 1e-5 relative."""
  import ra_ops as ops
  rng = np.random.RandomState(T + H)
  y_gt = np.zeros((B, T, H, W), np.float32)
  for b in range(B):
    for t in range(T - 1):  # the last instance stays empty (get_gt_box's fix-up path)
      y0, x0 = rng.randint(0, H - 8), rng.randint(0, W - 8)
      y_gt[b, t, y0:y0 + rng.randint(2, H - y0), x0:x0 + rng.randint(2, W - x0)] = 1
  params, box_gt = ops.gt_box(torch.tensor(y_gt, device=cuda), 0.2, 20.0)
  box = torch.tensor(rng.rand(B, H, W).astype(np.float32) ** 3, device=cuda)
  got = ops.box_iou_rects(box, params).cpu().numpy()
  ref = ops.pair_stats(box[:, None].contiguous(), box_gt, want=('iou_soft',))['iou_soft'].cpu().numpy().reshape(B, T)
  assert np.abs(got - ref).max() < 1e-5 * max(1e-3, np.abs(ref).max()), np.abs(got - ref).max()


def test_stacked_backward_equals_per_timestep_graph(cuda):
  """TrainStep.batched_backward (the T timesteps' backward passes as one stacked pass per layer: ConvStackFn over the
  [T, ...] slabs, ra_bn_act_pool_bwd_grouped_f32) against the per-timestep autograd graph on the same weights and
  knob draws: same loss pieces and matching, every parameter gradient within 2e-4 of the tensor's scale (the sums run
  in a different order), BatchNorm statistics identical; and the stacked path is the one that ran.  (Both forms with
  ra_ctrl_train_fwd_f32 as the sequential phase's controller: the decode loop's 16-workgroup controller the stacked step
  runs there by default sums in another order, and this random three-timestep network turns a 1e-7 difference in the
  first window into 2e-3 in a late BatchNorm gradient — two float32 trajectories, not one graph computed two ways.)"""
  import full_model
  opt, P, x, y_gt, s_gt = _case(wmul=0.6, **KNOB_OPT)
  res = {}
  for mode in (True, False):
    m = full_model.get_model(opt).load_weights(P)
    ts = ra_train.TrainStep(m)
    ts.batched_backward, ts.seq_ctrl_split = mode, False
    gen = torch.Generator(device='cuda').manual_seed(5)
    knobs = ts.draw_knobs(x.shape[0], gen)
    calls = {'n': 0}
    orig = ra_train.ConvStackFn.forward

    def counting(ctx, *a, _orig=orig):
      calls['n'] += 1
      return _orig(ctx, *a)
    ra_train.ConvStackFn.forward = staticmethod(counting)
    try:
      ts.bucket.zero_grad()
      loss, pieces, st = ts.forward_loss(x, y_gt, s_gt, knobs=knobs)
      loss.backward()
    finally:
      ra_train.ConvStackFn.forward = staticmethod(orig)
    torch.cuda.synchronize()
    assert calls['n'] == (21 if mode else 0), calls
    res[mode] = (float(loss.detach()), pieces['match'].cpu().numpy(), ts.bucket.grad.clone().cpu().numpy(), ts.stat.clone().cpu().numpy(),
                 {k: ts.bucket.grad_of[k].cpu().numpy().copy() for k in ts.bucket.names})
  a, b = res[True], res[False]
  assert abs(a[0] - b[0]) < 1e-6 * max(1.0, abs(b[0])) and (a[1] == b[1]).all()
  assert np.array_equal(a[3], b[3])
  gscale = max(np.abs(g).max() for g in b[4].values())
  for k, g in b[4].items():
    if _pre_bn_bias(k):  # a bias in front of BatchNorm has zero gradient: both are round-off
      assert np.abs(a[4][k]).max() < 2e-3 * gscale, k
      continue
    err = np.abs(a[4][k] - g).max() / max(np.abs(g).max(), 1e-3 * gscale)
    assert err < 2e-4, (k, err)


@pytest.mark.parametrize('case', ['kitti', 'cityscapes_iou_box_knob', 'cvppp_mse'])
def test_stacked_step_other_architectures_equal_per_timestep_graph(cuda, case):
  """The stacked step beyond the CVPPP architecture (round 4): skip connections into the decoder (concat(prev, skip) rebuilt
  over the stacked layer outputs), the 13 / 21 packed input channels of d_in / y_in, the 96-channel layer (grouped BatchNorm
  backward falls back to one call per timestep), use_knob + use_iou_box (the corner-IoU matrix rebuilt differentiably over
  all timesteps) and the 'mse' box loss (centre / log size of the stacked attention head): same loss pieces and matching
  as the per-timestep autograd graph, every parameter gradient within 2e-4 of the tensor's scale — and the stacked path ran."""
  import full_model
  H, W, T, B = 64, 128, 2, 2   # G = 8 glimpse cells: the fused controller's kernels take multiples of 4
  rng = np.random.RandomState(31)
  if case == 'cvppp_mse':
    opt = ora.make_opt('cvppp', H, W, T, base_learn_rate=1e-3, learn_rate_decay=0.96, steps_per_learn_rate_decay=5000, box_loss_fn='mse')
  else:
    over = dict(KNOB_OPT, use_iou_box=True) if case.startswith('cityscapes') else {}
    opt = ora.make_opt('cityscapes' if case.startswith('cityscapes') else 'kitti', H, W, T, base_learn_rate=1e-3, learn_rate_decay=0.96,
                       steps_per_learn_rate_decay=5000, **over)
  P = ora.random_params(opt, 32)
  for k in P:
    if ra_is_w(k):
      P[k] = (P[k] * 0.5).astype(np.float32)
  x = rng.rand(B, H, W, 3).astype(np.float32)
  extra = {}
  if opt.get('add_d_out', False):
    nsc = int(opt.get('num_semantic_classes', 1))
    extra = dict(d_in=np.eye(8, dtype=np.float32)[rng.randint(0, 8, (B, H, W))], y_in=ora.softmax(rng.randn(B, H, W, nsc)).astype(np.float32))
  y_gt, s_gt = np.zeros((B, T, H, W), np.float32), np.ones((B, T), np.float32)
  y_gt[:, 0, 6:30, 8:40] = 1
  y_gt[:, 1, 36:56, 50:120] = 1
  res = {}
  for mode in (True, False):
    m = full_model.get_model(opt).load_weights(P)
    ts = ra_train.TrainStep(m)
    ts.batched_backward, ts.seq_ctrl_split = mode, False
    assert ts._batched_ok([]) == mode
    knobs = ts.draw_knobs(B, torch.Generator(device='cuda').manual_seed(5)) if opt.get('use_knob', False) else None
    ts.bucket.zero_grad()
    loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt, knobs=knobs, **extra)
    loss.backward()
    torch.cuda.synchronize()
    assert ('ctrl_cnn_0_u' in ts._slabs) == mode
    res[mode] = ({k: float(pieces[k].detach()) for k in ('loss', 'box_loss', 'segm_loss', 'conf_loss')}, pieces['match'].cpu().numpy(),
                 pieces['match_box'].cpu().numpy(), {k: ts.bucket.grad_of[k].cpu().numpy().copy() for k in ts.bucket.names})
  a, b = res[True], res[False]
  for k in b[0]:
    assert abs(a[0][k] - b[0][k]) < 1e-5 * max(1.0, abs(b[0][k])), (k, a[0][k], b[0][k])
  assert (a[1] == b[1]).all() and (a[2] == b[2]).all()
  gscale = max(np.abs(g).max() for g in b[3].values())
  for k, g in b[3].items():
    if not _pre_bn_bias(k):
      err = np.abs(a[3][k] - g).max() / max(np.abs(g).max(), 1e-3 * gscale)
      assert err < 2e-4, (k, err)


def test_stacked_step_with_disable_overwrite_vs_oracle(cuda):
  """disable_overwrite = True (the reference's default, full_model.py:117-120: y *= 1 - canvas) through the stacked
  training graph: loss pieces and the whole gradient against float64 autograd (cosine; on this three-timestep random
  network single late-timestep BatchNorm parameters sit at 5-6 % from float64 in BOTH forms of the float32 graph), every
  parameter gradient within 2e-4 of the per-timestep graph, and the stacked form is what ran."""
  import full_model
  opt, P, x, y_gt, s_gt = _case(wmul=0.6, disable_overwrite=True)
  head, gref, stats = _oracle_grads(opt, P, x, y_gt, s_gt)
  grads = {}
  for mode in (True, False, 'split'):  # 'split': the stacked step as shipped, the decode loop's controller in its sequential phase
    m = full_model.get_model(opt).load_weights(P)
    ts = ra_train.TrainStep(m)
    ts.batched_backward, ts.seq_ctrl_split = bool(mode), mode == 'split'
    assert ts._batched_ok([]) == bool(mode)
    ts.bucket.zero_grad()
    loss, pieces, st = ts.forward_loss(x, y_gt, s_gt)
    loss.backward()
    assert ('ctrl_cnn_0_u' in ts._slabs) == bool(mode) and (getattr(ts, '_seqc', None) is not None) == (mode == 'split')
    for k in ('loss', 'iou_soft', 'iou_soft_box', 'conf_loss'):
      assert abs(float(pieces[k].detach()) - float(head[k])) < 2e-4 * max(1.0, abs(float(head[k]))), k
    assert (pieces['match'].cpu().numpy() == head['match'].numpy()).all()
    grads[mode] = {k: ts.bucket.grad_of[k].cpu().numpy().copy() for k in ts.bucket.names}
    cos = _grad_cosine(gref, lambda k: grads[mode][k], P, float(opt['weight_decay']))
    assert cos > 0.9995, (mode, cos)
  gscale = max(np.abs(g).max() for g in grads[False].values())
  for k, g in grads[False].items():
    if not _pre_bn_bias(k):
      err = np.abs(grads[True][k] - g).max() / max(np.abs(g).max(), 1e-3 * gscale)
      assert err < 2e-4, (k, err)


def test_failed_status_never_reaches_the_weights(cuda):
  """a step's solver / controller statuses are read by the host one step late (no sync per step), so the
  optimizer kernel itself must refuse the update of a step whose status word is non-zero (ra_adam_step_guarded_f32).  A
  negative status injected into the third step: weights, Adam moments and the step count after the raise are exactly those
  after step two — neither the failed step nor the one queued behind it has touched them — and flush_status raises when it is
  the last step that failed.  A controller time-out instead recovers: the step is skipped, the trainer moves to the
  one-workgroup controller and goes on."""
  import warnings
  import full_model
  from ra_native import RecAttendError
  opt, P, x, y_gt, s_gt = _case(T=2, wmul=0.6)
  feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': False}
  m = full_model.get_model(opt).load_weights(P)
  for _ in range(2):
    m.run(['loss', 'train_step'], feed)
  tr = m.trainer
  tr.flush_status()
  torch.cuda.synchronize()
  snap = (tr.bucket.param.clone(), tr.bucket.m.clone(), tr.bucket.v.clone(), tr.bucket.global_step)
  tr._inject_status = -3  # the third step's first matching "hit an inner cap" (hungarian.cc:126)
  m.run(['loss', 'train_step'], feed)   # step 3: issued, its update refused on the device; nobody has looked yet
  torch.cuda.synchronize()
  assert torch.equal(tr.bucket.param, snap[0]) and torch.equal(tr.bucket.m, snap[1]) and torch.equal(tr.bucket.v, snap[2])
  with pytest.raises(RecAttendError):
    m.run(['loss', 'train_step'], feed)  # step 4 notices step 3's record BEFORE its own optimizer launch is issued
  torch.cuda.synchronize()
  assert torch.equal(tr.bucket.param, snap[0]) and torch.equal(tr.bucket.m, snap[1]) and torch.equal(tr.bucket.v, snap[2])
  # ... and as the last step of a run: flush_status raises, the weights stand
  tr._status_pending = None
  tr._inject_status = -3
  m.run(['loss', 'train_step'], feed)
  with pytest.raises(RecAttendError):
    tr.flush_status()
  assert torch.equal(tr.bucket.param, snap[0])
  # a healthy step moves them again
  m.run(['loss', 'train_step'], feed)
  tr.flush_status()
  assert not torch.equal(tr.bucket.param, snap[0])
  # ---- a controller time-out (the sequential phase's 16-workgroup controller found a peer not resident)
  if getattr(tr, '_seqc', None) is not None and tr._seqc.get('ok'):
    torch.cuda.synchronize()
    snap2 = (tr.bucket.param.clone(), tr.bucket.global_step)
    orig = tr._seqc['status']
    tr._seqc['status'] = torch.ones_like(orig)  # what a timed-out launch leaves
    m.run(['loss', 'train_step'], feed)
    tr._seqc['status'] = orig
    torch.cuda.synchronize()
    assert torch.equal(tr.bucket.param, snap2[0])  # refused on the device
    with warnings.catch_warnings(record=True) as w:
      warnings.simplefilter('always')
      tr.flush_status()
    assert any('one-workgroup controller' in str(x_.message) for x_ in w)
    assert tr.seq_ctrl_split is False and tr.bucket.global_step == snap2[1] and tr.skipped_steps == 1
    loss, _ = m.run(['loss', 'train_step'], feed)  # goes on, on the one-workgroup controller
    tr.flush_status()
    assert np.isfinite(float(loss)) and not torch.equal(tr.bucket.param, snap2[0]) and tr.bucket.global_step == snap2[1] + 1


def test_controller_time_out_noticed_one_step_late(cuda):
  """full_model_train's loop never flushes, so a controller time-out of step k is noticed inside step k+1's run() —
  AFTER step k+1's graph has been launched on the 16-workgroup controller over the workspace step k left behind.  The check must
  not erase step k+1's own status (the words are snapshots, not aliases of the controller's sticky word), step k+1's update
  must be refused as well (a forced status word), both step numbers are given back, and the trainer goes on on the
  one-workgroup controller with a healthy step k+2."""
  import warnings
  import full_model
  opt, P, x, y_gt, s_gt = _case(T=2, wmul=0.6)
  feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': False}
  m = full_model.get_model(opt).load_weights(P)
  for _ in range(2):
    m.run(['loss', 'train_step'], feed)
  tr = m.trainer
  tr.flush_status()
  if getattr(tr, '_seqc', None) is None or not tr._seqc.get('ok'):
    pytest.skip('the sequential phase does not run the 16-workgroup controller here')
  torch.cuda.synchronize()
  snap = (tr.bucket.param.clone(), tr.bucket.m.clone(), tr.bucket.global_step)
  tr._inject_ctrl_status = 1  # step 3: "a controller workgroup timed out"
  m.run(['loss', 'train_step'], feed)
  assert tr._status_pending is not None and tr._status_pending[3] >= 1
  with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    m.run(['loss', 'train_step'], feed)  # step 4: launched, THEN step 3's record is read
  assert any('one-workgroup controller' in str(x_.message) for x_ in w)
  assert tr.seq_ctrl_split is False and tr._status_pending is not None and tr._status_pending[4] == 1  # step 4 carries the forced word
  torch.cuda.synchronize()
  assert torch.equal(tr.bucket.param, snap[0]) and torch.equal(tr.bucket.m, snap[1])  # neither step 3 nor step 4 reached the weights
  loss, _ = m.run(['loss', 'train_step'], feed)  # step 5 = the third step taken again, on the one-workgroup controller
  tr.flush_status()
  torch.cuda.synchronize()
  assert np.isfinite(float(loss)) and not torch.equal(tr.bucket.param, snap[0])
  assert tr.bucket.global_step == snap[2] + 1 and tr.skipped_steps == 2

