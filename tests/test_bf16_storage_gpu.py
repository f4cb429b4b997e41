"""bf16 STORAGE of the tensors between the conv layers' passes (model_opt['compute_dtype'] = 'bf16', the stacked training
step): every kernel that reads or writes a bf16 tensor against the SAME kernel on float32 copies of the same (bf16-exact)
values — the arithmetic is identical, so outputs agree exactly after the one rounding the bf16 form adds on its way out,
and float32 outputs (statistics, sums, parameter gradients) agree bit for bit."""
import ctypes as C

import numpy as np
import pytest
import torch

import ra_native as rn
import ra_ops as ops

pytestmark = pytest.mark.gpu


def _bf(a, cuda):
  """float32 array -> (bf16 tensor, float32 tensor holding the same bf16-exact values)."""
  t = torch.tensor(np.ascontiguousarray(a, dtype=np.float32), device=cuda).to(torch.bfloat16)
  return t, t.to(torch.float32)


def _f(*s, cuda):
  return torch.empty(s, dtype=torch.float32, device=cuda)


@pytest.mark.parametrize('cin,cout,pool_unused,ups', [(8, 16, 1, 0), (16, 32, 1, 0), (32, 16, 1, 1), (4, 8, 1, 0), (64, 64, 1, 0)])
def test_conv_bf16_storage_equals_operand_kernel(cuda, cin, cout, pool_unused, ups):
  rng = np.random.RandomState(cin + cout)
  B, Hs, Ws = 2, 12, 20
  xb, xf = _bf(rng.randn(B, Hs, Ws, cin), cuda)
  w = (rng.randn(3, 3, cin, cout) * 0.2).astype(np.float32)
  wp = torch.tensor(ops.pack_conv_weights(w), device=cuda)
  cp = ops.cout_padded(cout)
  sc, sh = torch.ones(cp, device=cuda), torch.tensor(rng.randn(cp).astype(np.float32) * 0.1, device=cuda)
  H, W = Hs * (1 + ups), Ws * (1 + ups)
  lib = rn.lib()
  npf = lib.ra_conv3x3_moments_part_floats(cout)
  ref, part0, n0 = _f(B, H, W, cout, cuda=cuda), _f(npf, cuda=cuda), C.c_int(0)
  rn.check(lib.ra_conv3x3_moments_f32(rn.ptr(xf), cin, None, 0, B, Hs, Ws, ups, rn.ptr(wp), rn.ptr(sc), rn.ptr(sh), cout, 0, 1, rn.ptr(ref),
                                      rn.ptr(part0), npf, C.byref(n0), rn.stream_ptr()), 'ref')
  for flags in (0, 1, 2, 3):
    xin = xb if flags & 1 else xf
    y = torch.empty((B, H, W, cout), dtype=torch.bfloat16 if flags & 2 else torch.float32, device=cuda)
    part, n1 = _f(npf, cuda=cuda), C.c_int(0)
    rn.check(lib.ra_conv3x3_bf16_f32(rn.ptr(xin), cin, None, 0, B, Hs, Ws, ups, rn.ptr(wp), rn.ptr(sc), rn.ptr(sh), cout, 0, 1, rn.ptr(y),
                                     rn.ptr(part), npf, C.byref(n1), flags, rn.stream_ptr()), 'bf16')
    want = ref.to(torch.bfloat16) if flags & 2 else ref
    assert torch.equal(y, want), flags
    assert n1.value == n0.value and torch.equal(part[:n1.value * cp * 4], part0[:n0.value * cp * 4])  # moments: the float32 accumulators
    # without moments, with pooling (the data-gradient / plain use)
    y2 = torch.empty((B, H, W, cout), dtype=torch.bfloat16 if flags & 2 else torch.float32, device=cuda)
    rn.check(lib.ra_conv3x3_bf16_f32(rn.ptr(xin), cin, None, 0, B, Hs, Ws, ups, rn.ptr(wp), rn.ptr(sc), rn.ptr(sh), cout, 0, 1, rn.ptr(y2),
                                     None, 0, None, flags, rn.stream_ptr()), 'bf16 plain')
    assert torch.equal(y2, want), flags


@pytest.mark.parametrize('C_,pool', [(8, 1), (8, 2), (32, 2), (64, 1)])
def test_bn_passes_bf16_storage_equal_float32_kernels(cuda, C_, pool):
  rng = np.random.RandomState(C_ + pool)
  G, B, H, W = 2, 3, 8, 12
  ub, uf = _bf(rng.randn(G * B, H, W, C_), cuda)
  dyb, dyf = _bf(rng.randn(G * B, H // pool, W // pool, C_), cuda)
  lib = rn.lib()
  par = [[torch.tensor(v.astype(np.float32), device=cuda) for v in (rng.randn(C_) * 0.1, rng.uniform(0.5, 1.5, C_), rng.uniform(0.5, 1.5, C_),
                                                                   rng.randn(C_) * 0.1)] for _ in range(G)]  # mean, var, gamma, beta
  # forward: normalise + ReLU + pool
  for g in range(G):
    mean, var, gamma, beta = par[g]
    sl = slice(g * B, (g + 1) * B)
    ref = _f(B, H // pool, W // pool, C_, cuda=cuda)
    rn.check(lib.ra_bn_act_pool_f32(rn.ptr(uf[sl]), rn.ptr(mean), rn.ptr(var), rn.ptr(gamma), rn.ptr(beta), C.c_float(1e-3), 1, pool, B, H, W,
                                    C_, rn.ptr(ref), rn.stream_ptr()), 'ref')
    for flags in (1, 3):
      y = torch.empty(ref.shape, dtype=torch.bfloat16 if flags & 2 else torch.float32, device=cuda)
      rn.check(lib.ra_bn_act_pool_bf16_f32(rn.ptr(ub[sl]), rn.ptr(mean), rn.ptr(var), rn.ptr(gamma), rn.ptr(beta), C.c_float(1e-3), 1, pool,
                                           B, H, W, C_, rn.ptr(y), flags, rn.stream_ptr()), 'bf16')
      assert torch.equal(y, ref.to(torch.bfloat16) if flags & 2 else ref), flags
  # backward, grouped (the stacked step) and per call
  def grouped(u, dy, du, flags):
    gg = [[torch.zeros(C_, device=cuda), torch.zeros(C_, device=cuda)] for _ in range(G)]
    cols = [[p[0] for p in par], [p[1] for p in par], [p[2] for p in par], [p[3] for p in par], [g_[0] for g_ in gg], [g_[1] for g_ in gg]]
    tab = torch.tensor([t.data_ptr() for col in cols for t in col], dtype=torch.int64).to(cuda)
    nbn = lib.ra_bn_workspace_floats(C_)
    ws, dgam, dbet = _f(G * nbn, cuda=cuda), _f(G, C_, cuda=cuda), _f(G, C_, cuda=cuda)
    if flags:
      rn.check(lib.ra_bn_act_pool_bwd_grouped_bf16_f32(rn.ptr(u), rn.ptr(dy), rn.ptr(tab), G, C.c_float(1e-3), 1, pool, B, H, W, C_, rn.ptr(ws),
                                                       ws.numel(), rn.ptr(dgam), rn.ptr(dbet), rn.ptr(du), flags, rn.stream_ptr()), 'g bf16')
    else:
      rn.check(lib.ra_bn_act_pool_bwd_grouped_f32(rn.ptr(u), rn.ptr(dy), rn.ptr(tab), G, C.c_float(1e-3), 1, pool, B, H, W, C_, rn.ptr(ws),
                                                  ws.numel(), rn.ptr(dgam), rn.ptr(dbet), rn.ptr(du), rn.stream_ptr()), 'g f32')
    torch.cuda.synchronize()
    return dgam.clone(), dbet.clone(), [[t.clone() for t in g_] for g_ in gg]
  du_ref = torch.empty_like(uf)
  r = grouped(uf, dyf, du_ref, 0)
  for flags, dy in ((3, dyb), (1, dyf)):
    du = torch.empty_like(ub)
    got = grouped(ub, dy, du, flags)
    assert torch.equal(du, du_ref.to(torch.bfloat16)), flags
    assert torch.equal(got[0], r[0]) and torch.equal(got[1], r[1])
    for a_, b_ in zip(got[2], r[2]):
      assert torch.equal(a_[0], b_[0]) and torch.equal(a_[1], b_[1])
  # per call, accumulating
  mean, var, gamma, beta = par[0]
  nbn = lib.ra_bn_workspace_floats(C_)
  def percall(u, dy, du, flags):
    ws, dg, db, ag, ab = _f(nbn, cuda=cuda), _f(C_, cuda=cuda), _f(C_, cuda=cuda), torch.zeros(C_, device=cuda), torch.zeros(C_, device=cuda)
    rn.check(lib.ra_bn_act_pool_bwd_acc_bf16_f32(rn.ptr(u), rn.ptr(dy), rn.ptr(mean), rn.ptr(var), rn.ptr(gamma), rn.ptr(beta), C.c_float(1e-3), 1,
                                                 pool, B, H, W, C_, rn.ptr(ws), ws.numel(), rn.ptr(dg), rn.ptr(db), rn.ptr(du), rn.ptr(ag), rn.ptr(ab),
                                                 flags, rn.stream_ptr()), 'acc')
    return dg, db, ag, ab
  d0 = torch.empty_like(uf[:B])
  r0 = percall(uf[:B].contiguous(), dyf[:B].contiguous(), d0, 0)
  d1 = torch.empty_like(ub[:B])
  r1 = percall(ub[:B].contiguous(), dyb[:B].contiguous(), d1, 3)
  assert torch.equal(d1, d0.to(torch.bfloat16)) and all(torch.equal(a_, b_) for a_, b_ in zip(r0, r1))


@pytest.mark.parametrize('cin,cout,ups', [(8, 16, 0), (16, 32, 0), (32, 16, 1), (4, 8, 0), (8, 8, 0), (64, 64, 0)])
def test_wgrad_bf16_storage_equals_operand_kernel(cuda, cin, cout, ups):
  rng = np.random.RandomState(cin * 3 + cout)
  B, Hs, Ws = 3, 8, 12
  H, W = Hs * (1 + ups), Ws * (1 + ups)
  xb, xf = _bf(rng.randn(B, Hs, Ws, cin), cuda)
  db_, df = _bf(rng.randn(B, H, W, cout), cuda)
  lib = rn.lib()
  nws = lib.ra_conv3x3_wgrad_workspace_floats(cin, cout, B, H, W)
  def run(x, du, fmt):
    ws, gw, gb = _f(nws, cuda=cuda), torch.zeros((3, 3, cin, cout), device=cuda), torch.zeros(cout, device=cuda)
    if fmt is None:
      rn.check(lib.ra_conv3x3_wgrad_acc_bf16ops_f32(rn.ptr(x), cin, B, Hs, Ws, ups, rn.ptr(du), cout, rn.ptr(ws), nws, None, cin, 0, rn.ptr(gw),
                                                    rn.ptr(gb), rn.stream_ptr()), 'ref')
    else:
      rn.check(lib.ra_conv3x3_wgrad_acc_bf16_f32(rn.ptr(x), cin, B, Hs, Ws, ups, rn.ptr(du), cout, rn.ptr(ws), nws, None, cin, 0, rn.ptr(gw),
                                                 rn.ptr(gb), fmt, rn.stream_ptr()), 'bf16')
    return gw, gb
  ref = run(xf, df, None)
  for fmt, x, du in ((3, xb, db_), (1, xb, df), (2, xf, db_), (0, xf, df)):
    got = run(x, du, fmt)
    if cout == 8 and not ups and ((cin == 8 and fmt == 3) or (cin == 4 and fmt == 2)):
      # round 5: these two storage combinations of the 8-output-channel layers run wgrad8b_kernel (K = 32 pixels per MFMA on the
      # bf16 tiles as they lie in memory): the same exact products in another summation order — float32 round-off, not bit equality
      for a_, b_ in zip(got, ref):
        assert float((a_ - b_).abs().max()) <= 2e-5 * max(1.0, float(b_.abs().max())), fmt
      continue
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), fmt


def test_subsample_odd_on_bf16_pixels(cuda):
  """The stride-2 data gradient keeps the odd positions: whole pixels are copied, so a bf16 tensor goes through the float32
  kernel as [.., C / 2] words."""
  rng = np.random.RandomState(2)
  B, Hs, Ws, Cc = 2, 6, 5, 8
  tb, tf = _bf(rng.randn(B, 2 * Hs, 2 * Ws, Cc), cuda)
  ref, got = _f(B, Hs, Ws, Cc, cuda=cuda), torch.empty((B, Hs, Ws, Cc), dtype=torch.bfloat16, device=cuda)
  lib = rn.lib()
  rn.check(lib.ra_subsample_odd_f32(rn.ptr(tf), B, Hs, Ws, Cc, rn.ptr(ref), rn.stream_ptr()), 'f32')
  rn.check(lib.ra_subsample_odd_f32(rn.ptr(tb), B, Hs, Ws, Cc // 2, rn.ptr(got), rn.stream_ptr()), 'bf16')
  assert torch.equal(got, ref.to(torch.bfloat16)) and torch.equal(ref, tf[:, 1::2, 1::2].contiguous())
