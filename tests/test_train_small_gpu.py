"""Round-4 launch cuts of the training step, each against what it replaces: the short-K parameter-gradient GEMM
(ra_gemm_tn_acc_f32) against float64 numpy, the one-workgroup BatchNorm passes of small tensors (the one-channel output
layer) against the formulas in float64, and the once-per-step gather that packs every filter (TrainStep._pack) against
the per-layer pack launches it replaces (bit for bit: packing only moves values)."""
import numpy as np
import pytest
import torch

import ra_native as rn

pytestmark = pytest.mark.gpu


def _t(a, cuda):
  return torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device=cuda)


@pytest.mark.parametrize('K,M,N,period,lda_pad', [(640, 320, 1024, 0, 7), (639, 256, 256, 5, 0), (128, 256, 9, 0, 3), (37, 20, 50, 4, 1)])
def test_gemm_tn_acc_plain(cuda, K, M, N, period, lda_pad):
  rng = np.random.RandomState(K + N)
  lda, ldb = M + lda_pad, N + 2
  A, B = rng.randn(K, lda), rng.randn(K, ldb)
  C0, b0 = rng.randn(M, N), rng.randn(N)
  keep = np.array([not (period and k % period == period - 1) for k in range(K)])
  want = C0 + A[keep][:, :M].T @ B[keep][:, :N]
  wantb = b0 + B[keep][:, :N].sum(axis=0)
  Ct, bt, At, Bt = _t(C0, cuda), _t(b0, cuda), _t(A, cuda), _t(B, cuda)
  rn.check(rn.lib().ra_gemm_tn_acc_f32(rn.ptr(At), lda, rn.ptr(Bt), ldb, K, M, N, period, rn.ptr(Ct), N, rn.ptr(bt), None, 0,
                                       0, rn.stream_ptr()), 'gemm')
  tol = 2e-5 * np.sqrt(K)
  assert np.abs(Ct.cpu().numpy() - want).max() < tol * max(1.0, np.abs(want).max())
  assert np.abs(bt.cpu().numpy() - wantb).max() < tol * max(1.0, np.abs(wantb).max())
  # without the bias row
  Ct2 = _t(C0, cuda)
  rn.check(rn.lib().ra_gemm_tn_acc_f32(rn.ptr(At), lda, rn.ptr(Bt), ldb, K, M, N, period, rn.ptr(Ct2), N, None, None, 0, 0,
                                       rn.stream_ptr()), 'gemm')
  assert torch.equal(Ct2, Ct)


def test_gemm_tn_acc_segmented_output(cuda):
  """The LSTM's layout: rows [0, Cf) / [Cf, Cf + hid) x four gate blocks -> eight weight tensors and four biases."""
  rng = np.random.RandomState(3)
  K, Cf, hid = 200, 32, 48
  M, N = Cf + hid, 4 * hid
  A, B = rng.randn(K, M + 5), rng.randn(K, N)
  full = A[:, :M].T @ B
  tops = [_t(rng.randn(Cf, hid), cuda) for _ in range(4)]
  bots = [_t(rng.randn(hid, hid), cuda) for _ in range(4)]
  bias = [_t(rng.randn(hid), cuda) for _ in range(4)]
  t0, b0, s0 = [t.cpu().numpy().copy() for t in tops], [t.cpu().numpy().copy() for t in bots], [t.cpu().numpy().copy() for t in bias]
  seg = torch.tensor([t.data_ptr() for t in tops + bots + bias], dtype=torch.int64, device=cuda)
  At, Bt = _t(A, cuda), _t(B, cuda)
  rn.check(rn.lib().ra_gemm_tn_acc_f32(rn.ptr(At), M + 5, rn.ptr(Bt), N, K, M, N, 0, None, 0, None, rn.ptr(seg), Cf, hid,
                                       rn.stream_ptr()), 'gemm seg')
  for j in range(4):
    cols = slice(j * hid, (j + 1) * hid)
    assert np.abs(tops[j].cpu().numpy() - (t0[j] + full[:Cf, cols])).max() < 5e-4
    assert np.abs(bots[j].cpu().numpy() - (b0[j] + full[Cf:, cols])).max() < 5e-4
    assert np.abs(bias[j].cpu().numpy() - (s0[j] + B[:, cols].sum(axis=0))).max() < 5e-4
  # a NULL entry is skipped, misaligned splits are refused
  seg2 = seg.clone()
  seg2[0] = 0
  before = tops[0].clone()
  rn.check(rn.lib().ra_gemm_tn_acc_f32(rn.ptr(At), M + 5, rn.ptr(Bt), N, K, M, N, 0, None, 0, None, rn.ptr(seg2), Cf, hid,
                                       rn.stream_ptr()), 'gemm seg')
  assert torch.equal(tops[0], before)
  assert rn.lib().ra_gemm_tn_acc_f32(rn.ptr(At), M + 5, rn.ptr(Bt), N, K, M, N, 0, None, 0, None, rn.ptr(seg), 24, hid,
                                     rn.stream_ptr()) == rn.RA_E_SHAPE


@pytest.mark.parametrize('C_,B,H,W', [(1, 8, 48, 48), (2, 3, 10, 12), (16, 2, 6, 6)])
def test_small_moments_one_launch(cuda, C_, B, H, W):
  rng = np.random.RandomState(C_)
  u = (rng.randn(B, H, W, C_) * 2.0 + 3.0).astype(np.float32)
  ut = _t(u, cuda)
  ws = torch.empty(rn.lib().ra_bn_workspace_floats(C_), dtype=torch.float32, device=cuda)
  mean, var = torch.empty(C_, device=cuda), torch.empty(C_, device=cuda)
  rn.check(rn.lib().ra_bn_moments_f32(rn.ptr(ut), B * H * W, C_, rn.ptr(ws), ws.numel(), rn.ptr(mean), rn.ptr(var), rn.stream_ptr()), 'moments')
  u64 = u.astype(np.float64).reshape(-1, C_)
  assert np.abs(mean.cpu().numpy() - u64.mean(axis=0)).max() < 2e-6 * 4
  assert np.abs(var.cpu().numpy() - u64.var(axis=0)).max() < 1e-5 * 4


def _bn_bwd_ref(u, dy, mean, var, gamma, beta, relu, pool, eps=1e-3):
  """float64: y = pool(relu(gamma * (u - mean) * rstd + beta)) with the batch statistics as functions of u."""
  ut = torch.tensor(u, dtype=torch.float64, requires_grad=True)
  g, b = torch.tensor(gamma, dtype=torch.float64, requires_grad=True), torch.tensor(beta, dtype=torch.float64, requires_grad=True)
  C_ = u.shape[-1]
  flat = ut.reshape(-1, C_)
  mu, vv = flat.mean(0), flat.var(0, unbiased=False)
  v = (ut - mu) / torch.sqrt(vv + eps) * g + b
  if relu:
    v = torch.relu(v)
  if pool == 2:
    v = torch.nn.functional.max_pool2d(v.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
  (v * torch.tensor(dy, dtype=torch.float64)).sum().backward()
  return ut.grad.numpy(), g.grad.numpy(), b.grad.numpy()


@pytest.mark.parametrize('C_,pool,relu', [(1, 1, 0), (1, 1, 1), (2, 2, 1)])
def test_small_bn_backward_one_workgroup_per_call_and_grouped(cuda, C_, pool, relu):
  import ra_train
  rng = np.random.RandomState(10 * C_ + pool + relu)
  G, B, H, W = 3, 4, 12, 16
  eps = float(ra_train.BN_EPS)
  lib = rn.lib()
  U = rng.randn(G, B, H, W, C_).astype(np.float32) * 1.5 + 0.3
  dY = rng.randn(G, B, H // pool, W // pool, C_).astype(np.float32)
  gam, bet = (1.0 + 0.2 * rng.randn(G, C_)).astype(np.float32), (0.2 * rng.randn(G, C_)).astype(np.float32)
  Ut, dYt = _t(U, cuda), _t(dY, cuda)
  means = [_t(U[g].reshape(-1, C_).astype(np.float64).mean(0), cuda) for g in range(G)]
  vars_ = [_t(U[g].reshape(-1, C_).astype(np.float64).var(0), cuda) for g in range(G)]
  gts, bts = [_t(gam[g], cuda) for g in range(G)], [_t(bet[g], cuda) for g in range(G)]
  accg, accb = [torch.zeros(C_, device=cuda) for _ in range(G)], [torch.zeros(C_, device=cuda) for _ in range(G)]
  nbn = lib.ra_bn_workspace_floats(C_)
  ws = torch.empty(G * nbn, device=cuda)
  # per call
  du1 = torch.empty_like(Ut)
  dg1, db1 = torch.empty(G, C_, device=cuda), torch.empty(G, C_, device=cuda)
  for g in range(G):
    rn.check(lib.ra_bn_act_pool_bwd_acc_f32(rn.ptr(Ut[g]), rn.ptr(dYt[g]), rn.ptr(means[g]), rn.ptr(vars_[g]), rn.ptr(gts[g]), rn.ptr(bts[g]),
                                            eps, relu, pool, B, H, W, C_, rn.ptr(ws), ws.numel(), rn.ptr(dg1[g]), rn.ptr(db1[g]), rn.ptr(du1[g]),
                                            rn.ptr(accg[g]), rn.ptr(accb[g]), rn.stream_ptr()), 'per call')
  for g in range(G):
    du, dgam, dbet = _bn_bwd_ref(U[g], dY[g], None, None, gam[g], bet[g], relu, pool, eps)
    assert np.abs(du1[g].cpu().numpy() - du).max() < 2e-5 * max(1.0, np.abs(du).max())
    assert np.abs(dg1[g].cpu().numpy() - dgam).max() < 1e-4 * max(1.0, np.abs(dgam).max())
    assert np.abs(db1[g].cpu().numpy() - dbet).max() < 1e-4 * max(1.0, np.abs(dbet).max())
    assert torch.equal(accg[g], dg1[g]) and torch.equal(accb[g], db1[g])
  # grouped: the same kernel, one workgroup per group of ONE launch -> the same bits
  tabs = torch.tensor([t.data_ptr() for t in means + vars_ + gts + bts + accg + accb], dtype=torch.int64, device=cuda)
  du2 = torch.empty_like(Ut)
  dg2, db2 = torch.empty(G, C_, device=cuda), torch.empty(G, C_, device=cuda)
  rn.check(lib.ra_bn_act_pool_bwd_grouped_f32(rn.ptr(Ut), rn.ptr(dYt), rn.ptr(tabs), G, eps, relu, pool, B, H, W, C_, rn.ptr(ws), ws.numel(),
                                              rn.ptr(dg2), rn.ptr(db2), rn.ptr(du2), rn.stream_ptr()), 'grouped')
  assert torch.equal(du2, du1) and torch.equal(dg2, dg1) and torch.equal(db2, db1)
  for g in range(G):
    assert torch.equal(accg[g], 2 * dg1[g]) and torch.equal(accb[g], 2 * db1[g])   # accumulated a second time


def test_prepacked_filters_equal_per_layer_packing(cuda):
  """Three optimisation steps with the once-per-step gather (steps 2 and 3 take every filter and padded bias from it)
  against the same steps with one pack launch per layer: identical losses and parameters, bit for bit."""
  import full_model
  import ra_train
  from test_train_gpu import _case
  opt, P, x, y_gt, s_gt = _case(T=2, B=2, wmul=0.6)
  res = {}
  for pre in (True, False):
    ra_train.TrainStep.prepack = pre
    try:
      m = full_model.get_model(opt).load_weights(P)
      losses = []
      for k in range(3):
        if getattr(m, 'trainer', None) is not None:
          m.trainer.use_graph = False
        else:
          ra_train.TrainStep.use_graph = False
        loss, _ = m.run(['loss', 'train_step'], {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': False})
        losses.append(float(loss))
      if pre:
        pk = m.trainer._pack
        assert pk.imap is not None and len(pk.entries) >= 20 and pk.filled == m.trainer._pack_epoch and not pk.pending
      res[pre] = (losses, m.state_dict_numpy())
    finally:
      ra_train.TrainStep.prepack = True
      ra_train.TrainStep.use_graph = True
  assert res[True][0] == res[False][0], (res[True][0], res[False][0])
  for k, v in res[False][1].items():
    assert np.array_equal(res[True][1][k], v), k


def test_controller_parameter_gradients_own_gemm_equals_library(cuda):
  """One optimisation step with the controller's end-of-backward products on ra_gemm_tn_acc_f32 against the same step on
  torch.mm / addmm / addmv (the library path it replaces): the updated parameters agree to float32 summation order."""
  import full_model
  import ra_train
  from test_train_gpu import _case
  opt, P, x, y_gt, s_gt = _case(T=3, B=2, wmul=0.6)
  res = {}
  for own in (True, False):
    ra_train.TrainStep.own_gemm, ra_train.TrainStep.use_graph = own, False
    try:
      m = full_model.get_model(opt).load_weights(P)
      loss, _ = m.run(['loss', 'train_step'], {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'aug': False})
      res[own] = (float(loss), {k: v.clone() for k, v in m.trainer.bucket.grad_of.items()})
    finally:
      ra_train.TrainStep.own_gemm, ra_train.TrainStep.use_graph = True, True
  assert res[True][0] == res[False][0]
  for k, v in res[False][1].items():
    a, b = res[True][1][k].cpu().numpy(), v.cpu().numpy()
    assert np.abs(a - b).max() <= 2e-5 * max(1e-3, np.abs(b).max()), (k, np.abs(a - b).max(), np.abs(b).max())


def test_pairwise_iou_reads_timestep_major_masks_in_place(cuda):
  """PairIoU(tmajor=True) on a [N,B,H,W] tensor against PairIoU on its [B,N,H,W] transposed copy: the same kernels through
  strides — IoU matrix and gradient bit for bit (ra_pair_stats_strided_f32 / ra_weighted_sum_multi_strided_f32)."""
  import ra_train
  rng = np.random.RandomState(0)
  N, B, M, H, W = 5, 3, 4, 16, 24
  a_t = torch.tensor(rng.rand(N, B, H, W).astype(np.float32), device=cuda, requires_grad=True)
  a_b = a_t.detach().transpose(0, 1).contiguous().requires_grad_(True)
  b = torch.tensor((rng.rand(B, M, H, W) > 0.6).astype(np.float32), device=cuda)
  g = torch.tensor(rng.randn(B, N, M).astype(np.float32), device=cuda)
  i_t = ra_train.PairIoU.apply(a_t, b, True)
  i_b = ra_train.PairIoU.apply(a_b, b)
  assert torch.equal(i_t, i_b)
  (i_t * g).sum().backward()
  (i_b * g).sum().backward()
  assert torch.equal(a_t.grad.transpose(0, 1), a_b.grad)


def test_stacked_step_reuses_the_sequential_phase_planes(cuda):
  """The stacked graph's box / patch / mask nodes take the planes the sequential phase wrote (no second forward of the
  extract and the two pastes): same loss and gradients as recomputing them — bit for bit when both phases run the same
  controller kernel."""
  import full_model
  import ra_train
  from test_train_gpu import _case, KNOB_OPT
  opt, P, x, y_gt, s_gt = _case(T=3, B=2, wmul=0.6, seed=5, **KNOB_OPT)
  res = {}
  for reuse in (True, False):
    m = full_model.get_model(opt).load_weights(P)
    ts = ra_train.TrainStep(m)
    ts.reuse_attn_planes, ts.seq_ctrl_split = reuse, False
    assert ts._batched_ok([])
    ts.bucket.zero_grad()
    rng = np.random.RandomState(5)
    B, T, H, W = 2, 3, 64, 64
    kd = {k: torch.tensor(v, dtype=torch.float32, device=cuda) for k, v in
          {'pad': rng.uniform(0.1, 0.3, (B, T, 1)), 'shift': rng.uniform(-0.05, 0.05, (B, T, 2)), 'u_box': rng.rand(B, T, 1),
           'u_segm': rng.rand(B, T, 1), 'segm_noise': 0.3 * rng.rand(T, B, H, W)}.items()}
    loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt, knobs=kd)
    loss.backward()
    res[reuse] = (float(loss), pieces['y_out'].detach().cpu().numpy().copy(), {k: v.cpu().numpy().copy() for k, v in ts.bucket.grad_of.items()})
  assert res[True][0] == res[False][0]
  assert res[True][1].shape == (2, 3, 64, 64) and np.array_equal(res[True][1], res[False][1])
  for k, v in res[False][2].items():
    assert np.array_equal(res[True][2][k], v), k


@pytest.mark.parametrize('batched', [True, False])
def test_loss_head_one_launch_equals_autograd_graph(cuda, batched):
  """LossHead (ra_loss_head_f32 / _bwd_f32: matched IoUs, confidence loss on the cumulative extrema, the two pairwise-IoU
  adjoints' coefficients, d s_out) against the graph of element-wise / scan / reduction ops it replaces: every loss piece
  and every gradient — stacked step (timestep-major masks) and per-timestep graph."""
  import full_model
  import ra_train
  from test_train_gpu import _case, KNOB_OPT
  opt, P, x, y_gt, s_gt = _case(T=3, B=2, wmul=0.6, seed=5, **(KNOB_OPT if batched else dict(stop_canvas_grad=False)))
  rng = np.random.RandomState(5)
  B, T, H, W = 2, 3, 64, 64
  kd = None
  if batched:
    kd = {k: torch.tensor(v, dtype=torch.float32, device=cuda) for k, v in
          {'pad': rng.uniform(0.1, 0.3, (B, T, 1)), 'shift': rng.uniform(-0.05, 0.05, (B, T, 2)), 'u_box': rng.rand(B, T, 1),
           'u_segm': rng.rand(B, T, 1), 'segm_noise': 0.3 * rng.rand(T, B, H, W)}.items()}
  res = {}
  for fused in (True, False):
    m = full_model.get_model(opt).load_weights(P)
    ts = ra_train.TrainStep(m)
    ts.fused_loss_head, ts.seq_ctrl_split = fused, False
    assert bool(ts._batched_ok([])) == batched
    ts.bucket.zero_grad()
    loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt, knobs=kd)
    loss.backward()
    res[fused] = ({k: float(pieces[k]) for k in ('loss', 'box_loss', 'segm_loss', 'conf_loss', 'iou_soft', 'iou_soft_box')},
                  pieces['match'].cpu().numpy().copy(), {k: v.cpu().numpy().copy() for k, v in ts.bucket.grad_of.items()})
  for k, v in res[False][0].items():
    assert abs(res[True][0][k] - v) < 2e-6 * max(1.0, abs(v)), (k, res[True][0][k], v)
  assert np.array_equal(res[True][1], res[False][1])
  scale = max(np.abs(g).max() for g in res[False][2].values())
  for k, g in res[False][2].items():
    assert np.abs(res[True][2][k] - g).max() < 2e-5 * max(np.abs(g).max(), 1e-3 * scale), (k, np.abs(res[True][2][k] - g).max(), np.abs(g).max())


def test_stacked_step_at_other_mlp_depths_vs_oracle(cuda):
  """The stacked training step with 3 glimpse-MLP and 2 controller-MLP layers (full_model.py:350-352,382-384; the run
  scripts use 2 and 1): the fused controller kernels and their per-layer parameter-gradient GEMMs against float64 autograd."""
  import full_model
  import ra_train
  from test_train_gpu import _case, _oracle_grads, _compare_grads
  opt, P, x, y_gt, s_gt = _case(T=2, B=2, wmul=0.6, seed=5, num_glimpse_mlp_layers=3, num_ctrl_mlp_layers=2, ctrl_mlp_dim=48)
  head, gref, _ = _oracle_grads(opt, P, x, y_gt, s_gt)
  m = full_model.get_model(opt).load_weights(P)
  ts = ra_train.TrainStep(m)
  assert ts._batched_ok([]) and (ts.d['n_gmlp'], ts.d['n_cmlp']) == (3, 2)
  ts.bucket.zero_grad()
  loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt)
  loss.backward()
  assert ts._ctl is not None and ts._ctl.n_g == 3 and ts._ctl.n_c == 2
  for k in ('loss', 'iou_soft', 'iou_soft_box', 'conf_loss'):
    assert abs(float(pieces[k]) - float(head[k])) < 3e-4 * max(1.0, abs(float(head[k]))), k
  _compare_grads(gref, lambda k: ts.bucket.grad_of[k].cpu().numpy(), P, float(opt['weight_decay']))


def test_knob_setup_fused_equals_elementwise(cuda):
  """ra_knob_setup_f32 (one launch on ra_gt_box_f32's partials) against the element-wise form of TrainStep._knob_setup
  (full_model.py:567-577,596-625): same float32 operations in the same order -> identical, including empty instances."""
  import ra_ops as ops
  import ra_train as rt
  rng = np.random.RandomState(7)
  B, T, H, W = 3, 6, 40, 56
  y = np.zeros((B, T, H, W), np.float32)
  for b in range(B):
    for t in range(T - 1 - b):  # the last instances of every image stay empty
      y0, x0 = rng.randint(0, H - 12), rng.randint(0, W - 12)
      y[b, t, y0:y0 + rng.randint(3, 12), x0:x0 + rng.randint(3, 12)] = 1.0
  y_gt = torch.tensor(y, device=cuda)
  u = lambda *s: torch.tensor(rng.rand(*s).astype(np.float32), device=cuda)
  knobs = {'pad': 0.1 + 0.2 * u(B, T, 1), 'shift': -0.05 + 0.1 * u(B, T, 2), 'u_box': u(B, T, 1), 'u_segm': u(B, T, 1)}

  class Stub:
    d = {'T': T}
    fused_knob_setup = True
    _sched = None

    class bucket:
      global_step = 700

  for timescale in (False, True):
    Stub.opt = dict(padding=16, knob_use_timescale=timescale, knob_base=1.0, knob_decay=0.9, steps_per_knob_decay=300,
                    knob_box_offset=300, knob_segm_offset=500)
    _, _, ws = ops.gt_box(y_gt, 0.25, 20.0, want_box=False, want_ws=True)
    fused = rt.TrainStep._knob_setup(Stub, y_gt, knobs, ws)
    plain = rt.TrainStep._knob_setup(Stub, y_gt, knobs, None)
    for a, b, name in zip(fused, plain, ('ctr', 'size', 'knob_box', 'knob_segm')):
      assert a.shape == b.shape and torch.equal(a, b), (name, timescale, float((a - b).abs().max()))
    assert 0 < float(fused[2].sum()) + float(fused[3].sum())  # the knobs are not all off at this step


def test_draw_knobs_in_place_equals_fresh(cuda):
  """The captured step's random draws written into its static buffers (the noise plane by uniform_(0, a)) against the
  fresh tensors of draw_knobs: same generator state -> the same bits (full_model.py:567-577,829-831)."""
  import ra_train as rt

  class Stub:
    d = {'T': 5, 'H': 24, 'W': 40}
    opt = dict(attn_box_padding_ratio=0.2, gt_box_pad_noise=0.1, gt_box_ctr_noise=0.05, gt_segm_noise=0.3)

    class bucket:
      param = torch.zeros(1, device=cuda)

  B = 3
  g = torch.Generator(device=cuda)
  g.manual_seed(99)
  fresh = rt.TrainStep.draw_knobs(Stub, B, g)
  shapes = rt.TrainStep.knob_shapes(Stub, B)
  assert {k: tuple(v.shape) for k, v in fresh.items()} == shapes
  out = {k: torch.full(s, -1.0, device=cuda) for k, s in shapes.items()}
  g.manual_seed(99)
  got = rt.TrainStep.draw_knobs(Stub, B, g, out=out)
  for k in fresh:
    assert torch.equal(got[k], fresh[k]), k
  assert float(fresh['segm_noise'].max()) < 0.3 and float(fresh['segm_noise'].min()) >= 0.0
