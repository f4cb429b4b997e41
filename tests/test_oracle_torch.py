"""The two CPU oracles against each other.  oracle/ra_oracle.py (NumPy) and
oracle/ra_oracle_torch.py (torch.nn.functional, differentiable) restate the reference's eval
forward and loss head independently; the neural path has no reference fixture ("parity unpinned"),
so their agreement to 1e-9 in float64 is the strongest pin available here.  The torch oracle's
autograd is then checked against finite differences of the NumPy oracle's loss: it is the gradient
reference for the training step's backward kernels (not built yet)."""
import numpy as np
import pytest
import torch

import ra_oracle as ora
import ra_oracle_torch as ort


def _inputs(opt, B, seed):
  rng = np.random.RandomState(seed)
  H, W = opt['inp_height'], opt['inp_width']
  x = rng.rand(B, H, W, 3)
  d_in = y_in = None
  if opt.get('add_d_out'):
    d_in = np.eye(8)[rng.randint(0, 8, (B, H, W))]
    y_in = ora.softmax(rng.randn(B, H, W, opt['num_semantic_classes']))
  return x, d_in, y_in


def _gt(rng, B, T, H, W):
  yy, xx = np.mgrid[0:H, 0:W]
  y_gt, s_gt = np.zeros((B, T, H, W)), np.zeros((B, T))
  for b in range(B):
    for t in range(min(T, 2 + b)):
      cy, cx, r = rng.randint(12, H - 12), rng.randint(12, W - 12), rng.randint(5, 11)
      y_gt[b, t] = ((yy - cy) ** 2 + (xx - cx) ** 2 < r * r)
      s_gt[b, t] = 1
  return y_gt, s_gt


@pytest.mark.parametrize('arch,H,W,T,over', [
    ('cvppp', 64, 64, 3, {}),
    ('cvppp', 64, 96, 2, dict(disable_overwrite=True, fixed_gamma=False, squash_ctrl_params=True)),
    ('kitti', 64, 96, 2, {}),
], ids=['cvppp', 'cvppp_flags', 'kitti_skip_dynamic_var'])
def test_two_restatements_agree(arch, H, W, T, over):
  opt = ora.make_opt(arch, H, W, T, **over)
  P = ora.random_params(opt, 21)
  x, d_in, y_in = _inputs(opt, 2, 22)
  ref = ora.full_model_forward(opt, P, x, d_in, y_in)
  with torch.no_grad():
    got, _ = ort.forward(opt, P, x, d_in, y_in)
  for k in ('y_out', 's_out', 'attn_box', 'attn_ctr', 'attn_size', 'x_patch', 'canvas'):
    assert np.abs(got[k].numpy() - ref[k]).max() < 1e-9, k
  if not over:
    assert ref['y_out'].max() > 0.5  # not a trivial all-sigmoid(-5) output


def test_loss_heads_agree_and_autograd_matches_finite_differences():
  # stop_canvas_grad off: finite differences see the path through the canvas, so must autograd
  opt = ora.make_opt('cvppp', 64, 64, 3, stop_canvas_grad=False)
  P = ora.random_params(opt, 16)
  rng = np.random.RandomState(3)
  x, _, _ = _inputs(opt, 2, 4)
  y_gt, s_gt = _gt(rng, 2, 3, 64, 64)
  ref = ora.loss_head(opt, ora.full_model_forward(opt, P, x), y_gt, s_gt)
  names = ('ctrl_mlp_b_0', 'score_mlp_b_0', 'attn_dcnn_b_6', 'ctrl_cnn_b_0', 'glimpse_mlp_b_1')
  fwd, Pt = ort.forward(opt, P, x, requires_grad=names)
  head = ort.loss_head(opt, fwd, y_gt, s_gt)
  for k in ('loss', 'iou_soft', 'iou_soft_box', 'conf_loss'):
    assert abs(float(head[k].detach()) - float(ref[k])) < 1e-9, k
  assert (head['match'].numpy() == ref['match']).all()
  head['loss'].backward()
  # central differences of the NumPy oracle's loss w.r.t. a few scalar parameters
  def loss_at(name, idx, delta):
    P2 = {k: v.copy() for k, v in P.items()}
    P2[name] = P2[name].astype(np.float64)
    P2[name].reshape(-1)[idx] += delta
    return float(ora.loss_head(opt, ora.full_model_forward(opt, P2, x), y_gt, s_gt)['loss'])
  checked = 0
  for name in names:
    g = Pt[name].grad.reshape(-1).numpy()
    idx = int(np.argmax(np.abs(g)))
    if abs(g[idx]) < 1e-8:
      continue
    eps = 1e-7  # the loss has kinks (max, ReLU, cummin): larger steps average across them
    fd = (loss_at(name, idx, eps) - loss_at(name, idx, -eps)) / (2 * eps)
    assert abs(fd - g[idx]) < 1e-4 * max(1.0, abs(g[idx])), (name, fd, g[idx])
    checked += 1
  assert checked >= 3
