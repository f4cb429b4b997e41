"""Parity of the loss / statistics head (csrc/ra_loss.hip, through the C ABI) against the
float64 NumPy oracle (oracle/ra_oracle.py: f_iou_pairwise ... loss_head, restating
full_model.py:913-1097 and modellib.py:28-37,71-155,265-339,382-415,482-511,663-701).
Tolerances: 2e-5 relative on the pairwise ratios (float32 sums of up to 2.6e5 terms against
float64), exact on the Hungarian matching, 2e-4 on the scalar statistics of a decoded batch."""
import numpy as np
import pytest
import torch

import ra_oracle as ora

pytestmark = pytest.mark.gpu


def dev(a, cuda):
  return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(cuda)


def synth_gt(rng, B, T, H, W, max_inst=None):
  """CVPPP-shaped ground truth: a few ellipses per image, sorted by area, s_gt = 1 for them."""
  yy, xx = np.mgrid[0:H, 0:W]
  y_gt = np.zeros((B, T, H, W), np.float32)
  s_gt = np.zeros((B, T), np.float32)
  for b in range(B):
    k = rng.randint(1, (max_inst or T) + 1)
    inst = []
    for _ in range(k):
      cy, cx = rng.uniform(0.15, 0.85) * H, rng.uniform(0.15, 0.85) * W
      ry, rx = rng.uniform(0.05, 0.2) * H, rng.uniform(0.05, 0.2) * W
      inst.append((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 < 1.0).astype(np.float32))
    inst.sort(key=lambda m: -m.sum())
    for t, m in enumerate(inst):
      y_gt[b, t] = m
      s_gt[b, t] = 1.0
  return y_gt, s_gt


@pytest.mark.parametrize('B,N,M,H,W', [(2, 5, 5, 32, 32), (3, 16, 16, 64, 64), (1, 20, 20, 48, 96),
                                       (2, 7, 21, 40, 36), (2, 21, 3, 20, 20), (1, 16, 16, 512, 512)])
def test_pair_stats(cuda, B, N, M, H, W):
  import ra_ops as ops
  rng = np.random.RandomState(N * 7 + M)
  a = rng.rand(B, N, H, W).astype(np.float32) ** 2
  b = (rng.rand(B, M, H, W) > 0.7).astype(np.float32)
  b[0, 0] = 0.0  # an empty instance
  a[0, N - 1] = 0.0
  st = ops.pair_stats(dev(a, cuda), dev(b, cuda))
  torch.cuda.synchronize()
  a64, b64 = a.astype(np.float64), b.astype(np.float64)
  ah = (a64 > 0.5).astype(np.float64)
  for key, ref in (('iou_soft', ora.f_iou_pairwise(a64, b64)), ('iou_hard', ora.f_iou_pairwise(ah, b64)),
                   ('dice_hard', ora.f_dice_pairwise(ah, b64)), ('sum_a', a64.sum(axis=(2, 3))),
                   ('sum_b', b64.sum(axis=(2, 3)))):
    got = st[key].cpu().numpy()
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), key


def test_pair_stats_is_deterministic_and_checks_arguments(cuda):
  import ra_ops as ops
  from ra_native import RecAttendError
  rng = np.random.RandomState(3)
  a, b = dev(rng.rand(2, 16, 64, 64), cuda), dev(rng.rand(2, 16, 64, 64), cuda)
  r1 = ops.pair_stats(a, b)['iou_soft'].clone()
  r2 = ops.pair_stats(a, b)['iou_soft']
  torch.cuda.synchronize()
  assert torch.equal(r1, r2)  # fixed-order reduction of the chunk partials
  with pytest.raises(RecAttendError):
    ops.pair_stats(dev(rng.rand(1, 33, 8, 8), cuda), dev(rng.rand(1, 4, 8, 8), cuda))  # N > 32
  with pytest.raises(RecAttendError):
    ops.pair_stats(dev(rng.rand(1, 2, 3, 5), cuda), dev(rng.rand(1, 2, 3, 5), cuda))  # HW % 4


@pytest.mark.parametrize('H,W,pad_ratio,min_pad', [(64, 64, 0.2, 20.0), (40, 72, 0.0, 10.0), (128, 128, 0.3, 4.0)])
def test_gt_box(cuda, H, W, pad_ratio, min_pad):
  import modellib
  rng = np.random.RandomState(H + W)
  y_gt, _ = synth_gt(rng, 3, 6, H, W)
  y_gt[1, 0] = 0.0
  y_gt[1, 0, 0:3, W - 4:W] = 1.0  # touches two borders
  y_gt[2, 1] *= 0.5              # soft ground truth: the reference's formula, not a bbox
  tl, br, box = ora.get_gt_box(y_gt.astype(np.float64), padding_ratio=pad_ratio, min_padding=min_pad)
  gtl, gbr, gbox = modellib.get_gt_box(dev(y_gt, cuda), padding_ratio=pad_ratio, min_padding=min_pad)
  torch.cuda.synchronize()
  assert np.abs(gtl.cpu().numpy() - tl).max() < 1e-4
  assert np.abs(gbr.cpu().numpy() - br).max() < 1e-4
  assert (gbox.cpu().numpy() == box).all()
  # get_gt_attn on top of it (modellib.py:644-660)
  ctr, size, lg_var, lg_gamma, box2, tl2, br2 = modellib.get_gt_attn(dev(y_gt, cuda), 48, 48, padding_ratio=pad_ratio,
                                                                    min_padding=min_pad)
  assert np.abs(ctr.cpu().numpy() - (tl + br) / 2).max() < 1e-4
  assert np.abs(size.cpu().numpy() - (br - tl)).max() < 1e-4
  # noisy ground-truth boxes as the training graph draws them (full_model.py:567-577): per-instance padding / centre shift tensors
  pr = (pad_ratio + rng.uniform(-0.1, 0.1, (3, 6, 1))).astype(np.float32)
  cs = rng.uniform(-0.05, 0.05, (3, 6, 2)).astype(np.float32)
  tl, br, box = ora.get_gt_box(y_gt.astype(np.float64), padding_ratio=pr.astype(np.float64), center_shift_ratio=cs.astype(np.float64),
                               min_padding=min_pad)
  gtl, gbr, gbox = modellib.get_gt_box(dev(y_gt, cuda), padding_ratio=dev(pr, cuda), center_shift_ratio=dev(cs, cuda), min_padding=min_pad)
  assert np.abs(gtl.cpu().numpy() - tl).max() < 1e-3 and np.abs(gbr.cpu().numpy() - br).max() < 1e-3
  assert (gbox.cpu().numpy() != box).mean() < 1e-3  # (a corner within round-off of a pixel index may fall either way)
  tl2, br2, box2 = modellib.get_gt_box(dev(y_gt, cuda), padding_ratio=0.2, center_shift_ratio=0.1, min_padding=min_pad)  # scalar shift
  rtl, rbr, rbox = ora.get_gt_box(y_gt.astype(np.float64), padding_ratio=0.2, center_shift_ratio=0.1, min_padding=min_pad)
  assert np.abs(tl2.cpu().numpy() - rtl).max() < 1e-3 and np.abs(br2.cpu().numpy() - rbr).max() < 1e-3
  assert (box2.cpu().numpy() != rbox).mean() < 1e-3


@pytest.mark.parametrize('B,T', [(4, 5), (3, 16), (2, 21)])
def test_segm_match_device_is_bit_exact(cuda, B, T):
  import ra_ops as ops
  rng = np.random.RandomState(B * 31 + T)
  iou = rng.rand(B, T, T).astype(np.float32) ** 3
  s_gt = (np.arange(T)[None, :] < rng.randint(1, T + 1, (B, 1))).astype(np.float32)
  ref = ora.f_segm_match(iou, s_gt)
  match, status = ops.segm_match(dev(iou, cuda), dev(s_gt, cuda))
  torch.cuda.synchronize()
  assert (status.cpu().numpy() == 0).all()
  assert (match.cpu().numpy() == ref).all()


def test_small_operators(cuda):
  """The [B,T]-sized reference-named operators of modellib against the oracle."""
  import modellib
  rng = np.random.RandomState(5)
  B, T, H, W = 3, 6, 32, 32
  y_gt, s_gt = synth_gt(rng, B, T, H, W)
  a = rng.rand(B, T, H, W).astype(np.float32)
  a64, g64 = a.astype(np.float64), y_gt.astype(np.float64)
  da, dg, ds = dev(a, cuda), dev(y_gt, cuda), dev(s_gt, cuda)
  iou = ora.f_iou_pairwise(a64, g64)
  close = lambda got, ref, tol=2e-5: np.abs(got.cpu().numpy() - ref).max() <= tol * max(1.0, np.abs(ref).max())
  assert close(modellib.f_iou(da, dg, T, pairwise=True), iou)
  assert close(modellib.f_iou(da, dg), np.einsum('bii->bi', iou))
  assert close(modellib.f_dice(da, dg, T, pairwise=True), ora.f_dice_pairwise(a64, g64))
  assert close(modellib.f_inter(da, dg), (a64 * g64).sum(axis=(2, 3)), 1e-4)
  assert close(modellib.f_union(da, dg), (a64 + g64 - a64 * g64 + 1e-5).sum(axis=(2, 3)), 1e-4)
  diou = dev(iou, cuda)
  assert close(modellib.f_weighted_coverage(diou, dg), ora.f_weighted_coverage(iou, g64))
  ident = ora.get_identity_match(s_gt.astype(np.float64))
  assert close(modellib.get_identity_match(B, T, ds), ident)
  cnt = np.maximum(1.0, ident.sum(axis=(1, 2)))
  assert close(modellib.f_unweighted_coverage(diou, dev(cnt, cuda)), ora.f_unweighted_coverage(iou, cnt))
  s_out = rng.rand(B, T)
  assert close(modellib.f_conf_loss(dev(s_out, cuda), dev(ident, cuda), T), ora.f_conf_loss(s_out, ident))
  acc, dic, dica = ora.f_count_stats(s_out, s_gt.astype(np.float64))
  assert close(modellib.f_count_acc(dev(s_out, cuda), ds), acc)
  assert close(modellib.f_dic(dev(s_out, cuda), ds), dic)
  assert close(modellib.f_dic(dev(s_out, cuda), ds, abs=True), dica)
  m = ora.f_greedy_match(iou[:, 0], np.zeros((B, T)))
  assert close(modellib.f_greedy_match(dev(iou[:, 0], cuda), torch.zeros((B, T), device=cuda)), m)
  with pytest.raises(NotImplementedError):
    modellib.f_match_loss(da, dg, None, T, None)


SCALARS = ('loss', 'box_loss', 'segm_loss', 'conf_loss', 'iou_soft', 'wt_cov_soft', 'unwt_cov_soft',
           'iou_hard', 'wt_cov_hard', 'unwt_cov_hard', 'dice', 'count_acc', 'dic', 'dic_abs')


@pytest.mark.parametrize('over', [{}, {'fixed_order': True}, {'segm_loss_fn': 'wt_cov'}],
                         ids=['matching', 'fixed_order', 'wt_cov'])
def test_loss_head_of_a_decoded_batch(cuda, over):
  """Model.run(['loss', 'iou_soft', ...]) = decode + loss head, against the oracle's decode +
  loss_head on the same weights, inputs and synthetic ground truth."""
  import full_model
  opt = ora.make_opt('cvppp', 128, 128, 5, **over)
  P = ora.random_params(opt, 16)
  rng = np.random.RandomState(17)
  B = 2
  x = rng.rand(B, 128, 128, 3).astype(np.float32)
  y_gt, s_gt = synth_gt(rng, B, 5, 128, 128, max_inst=4)
  ref = ora.loss_head(opt, ora.full_model_forward(opt, P, x), y_gt, s_gt)
  m = full_model.get_model(opt).load_weights(P)
  names = list(SCALARS) + ['match', 'match_box', 'attn_box_gt', 'y_out']
  feed = {'x': x, 'phase_train': False, 'y_gt': y_gt, 's_gt': s_gt}
  for rep in range(2):  # second call replays the decode graph
    out = dict(zip(names, m.run(names, feed, as_numpy=True)))
    assert (out['attn_box_gt'] == ref['attn_box_gt']).all()
    assert (out['match'] == ref['match']).all() and (out['match_box'] == ref['match_box']).all()
    for k in SCALARS:
      assert abs(float(out[k]) - float(ref[k])) < 2e-4 * max(1.0, abs(float(ref[k]))), k
    assert all(int(s.sum()) == 0 for s in m.match_status)
  with pytest.raises(Exception):
    m.run(['loss'], {'x': x, 'phase_train': False})  # no ground truth in the feed
  loss, _ = m.run(['loss', 'train_step'], feed)  # the same feed drives one optimizer step (tests/test_train_gpu.py)
  assert np.isfinite(float(loss)) and float(m['global_step']) == 1.0


def test_loss_head_full_size_properties(cuda):
  """cfg2 shapes (512x512, T=16): size-independent properties of the streaming pass —
  IoU of a set with itself is 1 - eps*HW/|a|-ish, disjoint instances have IoU 0, the matching
  of y_out = y_gt is the identity, and every statistic is then its ideal value."""
  import ra_ops as ops
  rng = np.random.RandomState(9)
  B, T, H, W = 2, 16, 512, 512
  y_gt, s_gt = synth_gt(rng, B, T, H, W, max_inst=10)
  g = dev(y_gt, cuda)
  st = ops.pair_stats(g, g)
  s = dev(s_gt, cuda)
  match, status = ops.segm_match(st['iou_soft'], s)
  torch.cuda.synchronize()
  iou = st['iou_soft'].cpu().numpy()
  area = y_gt.sum(axis=(2, 3))
  for b in range(B):
    for t in range(T):
      if s_gt[b, t]:
        assert abs(iou[b, t, t] - area[b, t] / (area[b, t] + 1e-5 * H * W)) < 1e-5
  assert (iou <= 1.0 + 1e-6).all() and (iou >= 0).all()
  assert np.abs(iou - np.transpose(iou, (0, 2, 1))).max() < 1e-6  # symmetric for a == b
  ident = ora.get_identity_match(s_gt)
  # overlapping ellipses may legitimately match off the diagonal only if that has higher total IoU;
  # with y_out == y_gt the diagonal (IoU ~1 each) is optimal
  assert (match.cpu().numpy() == ident).all() and (status.cpu().numpy() == 0).all()
  stats = ops.loss_stats(st['iou_soft'], st['iou_hard'], st['dice_hard'], match, st['iou_soft'], match,
                         s, s, st['sum_b']).cpu().numpy()
  named = dict(zip(ops.STAT_NAMES, stats))
  assert named['iou_soft'] > 0.99 and named['iou_hard'] > 0.99 and named['dice'] > 0.99
  assert named['wt_cov_hard'] > 0.99 and named['count_acc'] == 1.0 and named['dic'] == 0.0


def _load_loss_fixture():
  import os
  fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'loss_head_cvppp_128.npz'),
               allow_pickle=True)
  shape = tuple(int(v) for v in fx['y_gt_shape'])
  y_gt = np.unpackbits(fx['y_gt_bits'])[:int(np.prod(shape))].reshape(shape).astype(np.float32)
  return fx, y_gt


def test_loss_head_golden_fixture(cuda):
  """The committed vector tests/golden/loss_head_cvppp_128.npz (make_fixtures.py)."""
  import full_model
  fx, y_gt = _load_loss_fixture()
  opt = fx['opt'].item()
  m = full_model.get_model(opt).load_weights(ora.random_params(opt, int(fx['seed'])))
  names = list(SCALARS) + ['match', 'match_box']
  out = dict(zip(names, m.run(names, {'x': fx['x'], 'phase_train': False, 'y_gt': y_gt, 's_gt': fx['s_gt']},
                              as_numpy=True)))
  assert (out['match'] == fx['match']).all() and (out['match_box'] == fx['match_box']).all()
  for k in SCALARS:
    assert abs(float(out[k]) - float(fx[k])) < 2e-4 * max(1.0, abs(float(fx[k]))), k
