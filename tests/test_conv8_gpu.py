"""The bf16 mode's eight-output-channel convolution (csrc/ra_conv8.hip: the tile lives in LDS as bf16, one ds_read_b128 per
MFMA operand) against a float64 convolution of the SAME bf16-rounded operands — the contraction K1's bf16 kernels compute
(full_model.py:240-262's first CNN layers at full resolution and their data gradients).  Launches of at least 64*64*8 pixels
take this kernel; smaller ones stay on K1 (tests/test_bf16_storage_gpu.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

import ra_native as rn
import ra_ops as ops

pytestmark = pytest.mark.gpu


def _ref(x64, w, sh, relu):
  y = torch.nn.functional.conv2d(x64.permute(0, 3, 1, 2), torch.tensor(w, dtype=torch.float64, device=x64.device).permute(3, 2, 0, 1), padding=1)
  y = y.permute(0, 2, 3, 1) + sh.double()[:y.shape[1]]
  return torch.clamp_min(y, 0.0) if relu else y


@pytest.mark.parametrize('cin,in_bf,cout', [(4, 0, 8), (8, 0, 8), (8, 1, 8), (16, 1, 8), (8, 0, 16), (8, 1, 16), (16, 1, 16)])
@pytest.mark.parametrize('relu', [0, 1])
def test_conv8_vs_float64(cuda, cin, in_bf, cout, relu):
  rng = np.random.RandomState(10 * cin + in_bf + cout)
  B, H, W = 3, 77, 150  # ragged against the 8 x 64 tile
  x = torch.tensor(rng.randn(B, H, W, cin).astype(np.float32), device=cuda)
  xb = x.to(torch.bfloat16)
  xin = xb if in_bf else x
  w = (rng.randn(3, 3, cin, cout) * 0.2).astype(np.float32)
  wb = torch.tensor(w).to(torch.bfloat16).double().numpy()
  wp = torch.tensor(ops.pack_conv_weights(w), device=cuda)
  cp = ops.cout_padded(cout)
  sc, sh = torch.ones(cp, device=cuda), torch.tensor(rng.randn(cp).astype(np.float32) * 0.1, device=cuda)
  want = _ref(xb.double(), wb, sh, relu)
  want_pre = _ref(xb.double(), wb, sh, 0)
  lib = rn.lib()
  npf = lib.ra_conv3x3_moments_part_floats(cout)
  for out_bf in (0, 1):
    for mom in (0, 1):
      y = torch.full((B, H, W, cout), 7.0, dtype=torch.bfloat16 if out_bf else torch.float32, device=cuda)
      part, n1 = torch.zeros(npf, device=cuda), C.c_int(0)
      rn.check(lib.ra_conv3x3_bf16_f32(rn.ptr(xin), cin, None, 0, B, H, W, 0, rn.ptr(wp), rn.ptr(sc), rn.ptr(sh), cout, relu, 1, rn.ptr(y),
                                       rn.ptr(part) if mom else None, npf if mom else 0, C.byref(n1) if mom else None, in_bf | 2 * out_bf,
                                       rn.stream_ptr()), 'conv8')
      err = (y.double() - want).abs()
      tol = (2.0 ** -8) * want.abs() + 1e-5 if out_bf else 2e-5 * (1.0 + want.abs())
      assert bool((err <= tol).all()), (out_bf, mom, float(err.max()))
      if mom:  # tf.nn.moments of the pre-activation output, from the float32 accumulators
        mean, var = torch.empty(cout, device=cuda), torch.empty(cout, device=cuda)
        rn.check(lib.ra_bn_moments_from_partials_f32(rn.ptr(part), n1.value, cout, rn.ptr(mean), rn.ptr(var), rn.stream_ptr()), 'moments')
        flat = want_pre.reshape(-1, cout)
        np.testing.assert_allclose(mean.cpu().numpy(), flat.mean(0).cpu().numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(var.cpu().numpy(), flat.var(0, unbiased=False).cpu().numpy(), rtol=1e-5)


@pytest.mark.parametrize('cin,cout', [(4, 8), (8, 8), (8, 16), (16, 16)])
@pytest.mark.parametrize('relu', [0, 1])
def test_conv8_float32_vs_float64(cuda, cin, cout, relu):
  """float32 mode: 8 output channels on the 16-block MFMA form (v_mfma_f32_4x4x1_16B_f32, no padded output channels), 16 on
  conv16_kernel (16x16x4 with one ds_read_b128 per four MFMAs) — exact float32 products, compared with a float64
  convolution of the same float32 operands."""
  rng = np.random.RandomState(100 + cin + cout)
  B, H, W = 3, 77, 150
  x = torch.tensor(rng.randn(B, H, W, cin).astype(np.float32), device=cuda)
  w = (rng.randn(3, 3, cin, cout) * 0.2).astype(np.float32)
  wp = torch.tensor(ops.pack_conv_weights(w), device=cuda)
  cp = ops.cout_padded(cout)
  sc, sh = torch.ones(cp, device=cuda), torch.tensor(rng.randn(cp).astype(np.float32) * 0.1, device=cuda)
  want = _ref(x.double(), w.astype(np.float64), sh, relu)
  want_pre = _ref(x.double(), w.astype(np.float64), sh, 0)
  lib = rn.lib()
  npf = lib.ra_conv3x3_moments_part_floats(cout)
  y = torch.full((B, H, W, cout), 7.0, device=cuda)
  rn.check(lib.ra_conv3x3_f32(rn.ptr(x), cin, None, 0, B, H, W, 0, rn.ptr(wp), rn.ptr(sc), rn.ptr(sh), cout, relu, 1, None, -1, rn.ptr(y),
                              rn.stream_ptr()), 'plain')
  assert float((y.double() - want).abs().max()) < 2e-5
  y2 = torch.full((B, H, W, cout), 7.0, device=cuda)
  part, n1 = torch.zeros(npf, device=cuda), C.c_int(0)
  rn.check(lib.ra_conv3x3_moments_f32(rn.ptr(x), cin, None, 0, B, H, W, 0, rn.ptr(wp), rn.ptr(sc), rn.ptr(sh), cout, relu, 0, rn.ptr(y2),
                                      rn.ptr(part), npf, C.byref(n1), rn.stream_ptr()), 'moments')
  assert torch.equal(y2, y)
  mean, var = torch.empty(cout, device=cuda), torch.empty(cout, device=cuda)
  rn.check(lib.ra_bn_moments_from_partials_f32(rn.ptr(part), n1.value, cout, rn.ptr(mean), rn.ptr(var), rn.stream_ptr()), 'moments')
  flat = want_pre.reshape(-1, cout)
  np.testing.assert_allclose(mean.cpu().numpy(), flat.mean(0).cpu().numpy(), rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(var.cpu().numpy(), flat.var(0, unbiased=False).cpu().numpy(), rtol=1e-5)


# ---- the filter gradient of the same layers (csrc/ra_train.hip wgrad8_kernel: Cout = 8, Cin in {4, 8}, float32, both tiles
# HBM -> LDS directly; wgrad_kernel for the other channel counts) on shapes ragged against its 8 x 32 tiles
def _wgrad_ref(x, du):
  """dW[ky,kx,ci,co] = sum over pixels of x[.. + tap, ci] * du[.., co] (SAME padding), db = sum du, in float64 on the device."""
  xi = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))
  B, H, W, Co = du.shape
  d = du.double()
  dw = torch.stack([torch.stack([torch.einsum('bhwc,bhwd->cd', xi[:, ky:ky + H, kx:kx + W], d) for kx in range(3)]) for ky in range(3)])
  return dw, d.sum(dim=(0, 1, 2))


def _rel(a, b):
  return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize('cin,cout', [(4, 8), (8, 8), (8, 16), (16, 32), (12, 8)])
@pytest.mark.parametrize('B,H,W', [(3, 77, 150), (1, 9, 33), (2, 8, 32)])
def test_wgrad_ragged_vs_float64(cuda, cin, cout, B, H, W):
  """ra_conv3x3_wgrad_f32 on ragged multi-tile shapes (wgrad8_kernel<4> / <8> for the two 8-output-channel forms; the
  generic wgrad_kernel beside them), against a float64 correlation of the same float32 operands."""
  rng = np.random.RandomState(cin * 100 + cout + H)
  x = torch.tensor(rng.randn(B, H, W, cin).astype(np.float32), device=cuda)
  du = torch.tensor(rng.randn(B, H, W, cout).astype(np.float32), device=cuda)
  lib = rn.lib()
  nws = lib.ra_conv3x3_wgrad_workspace_floats(cin, cout, B, H, W)
  ws = torch.full((nws,), float('nan'), device=cuda)
  dw, db = torch.full((3, 3, cin, cout), 7.0, device=cuda), torch.full((cout,), 7.0, device=cuda)
  rn.check(lib.ra_conv3x3_wgrad_f32(rn.ptr(x), cin, B, H, W, 0, rn.ptr(du), cout, rn.ptr(ws), nws, rn.ptr(dw), rn.ptr(db), rn.stream_ptr()),
           'wgrad')
  rw, rb = _wgrad_ref(x, du)
  assert _rel(dw, rw) < 2e-5 and _rel(db, rb) < 2e-5, (_rel(dw, rw), _rel(db, rb))


@pytest.mark.parametrize('cin', [4, 8])
def test_wgrad8_pointer_table_vs_float64(cuda, cin):
  """ra_conv3x3_wgrad_multi_acc_f32 — the stacked step's call: the layer's T calls through two device pointer tables
  (xtab / dutab), the sums ADDED to gw / gb in the reference layout — for the 8-output-channel layers on a ragged shape,
  segments of 2 images at unrelated addresses."""
  rng = np.random.RandomState(cin)
  nseg, Bseg, H, W, cout = 3, 2, 45, 70, 8
  xs = [torch.tensor(rng.randn(Bseg, H, W, cin).astype(np.float32), device=cuda) for _ in range(nseg)]
  junk = [torch.full((1000 + 77 * i,), float('nan'), device=cuda) for i in range(nseg)]  # keeps the segments apart
  dus = [torch.tensor(rng.randn(Bseg, H, W, cout).astype(np.float32), device=cuda) for _ in range(nseg)]
  lib = rn.lib()
  tab = torch.zeros(128, dtype=torch.int64, device=cuda)
  for off, ts in ((0, xs), (64, dus)):
    host = (C.c_void_p * nseg)(*[t.data_ptr() for t in ts])
    rn.check(lib.ra_ptr_table(host, nseg, tab.data_ptr() + 8 * off, rn.stream_ptr()), 'ptr_table')
  nws = lib.ra_conv3x3_wgrad_workspace_floats(cin, cout, nseg * Bseg, H, W)
  ws = torch.full((nws,), float('nan'), device=cuda)
  g0w, g0b = rng.randn(3, 3, cin, cout).astype(np.float32), rng.randn(cout).astype(np.float32)
  gw, gb = torch.tensor(g0w, device=cuda), torch.tensor(g0b, device=cuda)
  rn.check(lib.ra_conv3x3_wgrad_multi_acc_f32(tab.data_ptr(), tab.data_ptr() + 8 * 64, nseg, cin, Bseg, H, W, 0, cout, rn.ptr(ws), nws,
                                              None, cin, 0, rn.ptr(gw), rn.ptr(gb), 0, rn.stream_ptr()), 'wgrad_multi')
  rw, rb = _wgrad_ref(torch.cat(xs), torch.cat(dus))
  rw, rb = rw + torch.tensor(g0w, device=cuda).double(), rb + torch.tensor(g0b, device=cuda).double()
  assert _rel(gw, rw) < 2e-5 and _rel(gb, rb) < 2e-5, (_rel(gw, rw), _rel(gb, rb))
  del junk


def test_full_resolution_layer_gradient_pin(cuda):
  """ONE layer at cfg4's resolution, where conditioning is no excuse: the controller CNN's 8 -> 8 layer (L1: conv + b ->
  BatchNorm on batch moments -> ReLU -> 2x2 max-pool, nnlib.py:229-253 with :98-112) at 512 x 512, B = 8, through
  ra_train.ConvBNActPool — data gradient (the MFMA conv on the transposed packing), filter gradient (wgrad8_kernel), BatchNorm
  backward — against float64 autograd on the device.  Bars: 1e-4 relative (max-abs) for dW, dgamma, dbeta, the output and the
  statistics; for dx 1e-4 in L2 and at most 1e-5 of the elements off by more than 1e-4 of the scale (a max-pool arg-max or a
  ReLU sign decided differently in float32 moves single pixels by O(dy), not the sums)."""
  import ra_train
  rng = np.random.RandomState(2)
  B, H, W, cin, cout = 8, 512, 512, 8, 8
  x = torch.tensor(rng.randn(B, H, W, cin).astype(np.float32), device=cuda)
  w = torch.tensor((rng.randn(3, 3, cin, cout) * 0.2).astype(np.float32), device=cuda)
  b = torch.tensor((rng.randn(cout) * 0.1).astype(np.float32), device=cuda)
  gam = torch.tensor(rng.uniform(0.5, 1.5, cout).astype(np.float32), device=cuda)
  bet = torch.tensor((rng.randn(cout) * 0.2).astype(np.float32), device=cuda)
  dy = torch.tensor(rng.randn(B, H // 2, W // 2, cout).astype(np.float32), device=cuda)
  # float64 reference, on the device (torch = plumbing; the arithmetic under test is the HIP kernels')
  xr, wr, br, gr, ber = [t.double().requires_grad_(True) for t in (x, w, b, gam, bet)]
  u = torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), wr.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1) + br
  mean = u.mean(dim=(0, 1, 2))
  var = ((u - mean) ** 2).mean(dim=(0, 1, 2))
  v = torch.relu((u - mean) * torch.rsqrt(var + 1e-3) * gr + ber)
  yr = torch.nn.functional.max_pool2d(v.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
  (yr * dy.double()).sum().backward()
  del u, v
  xd, wd, bd, gd, bed = [t.clone().requires_grad_(True) for t in (x, w, b, gam, bet)]
  meta = dict(transposed=False, stride=1, pool=2, relu=True, chan_map=None)
  yd, md, vd = ra_train.ConvBNActPool.apply(xd, wd, bd, gd, bed, meta)
  (yd * dy).sum().backward()
  assert _rel(yd.detach(), yr.detach()) < 1e-4
  assert _rel(md, mean.detach()) < 1e-4 and _rel(vd, var.detach()) < 1e-4
  errs = {n: _rel(a.grad, r.grad) for n, a, r in (('dw', wd, wr), ('dgamma', gd, gr), ('dbeta', bed, ber))}
  assert max(errs.values()) < 1e-4, errs
  diff = (xd.grad.double() - xr.grad)
  scale = float(xr.grad.abs().max())
  l2 = float(diff.norm() / xr.grad.norm())
  frac = float((diff.abs() > 1e-4 * scale).double().mean())
  print('8->8 layer at 512x512, B=8: dW %.1e dgamma %.1e dbeta %.1e | dx L2 %.1e, outliers %.1e' % (
      errs['dw'], errs['dgamma'], errs['dbeta'], l2, frac))
  assert l2 < 1e-4 and frac < 1e-5, (l2, frac)


@pytest.mark.parametrize('cin', [4, 8])
@pytest.mark.parametrize('table', [False, True])
def test_wgrad8_bf16_storage_vs_float64(cuda, cin, table):
  """wgrad8b_kernel (round 5): the bf16 mode's filter gradient of the two full-resolution layers — dU stored as bf16, x stored as
  bf16 (8 -> 8) or the float32 packed image (4 -> 8, rounded to bf16 as an operand) — on v_mfma_f32_16x16x32_bf16 with K = 32
  pixels, through ra_conv3x3_wgrad_acc_bf16_f32 and, with `table`, the stacked step's pointer-table call.  Against a float64
  correlation of the SAME bf16 values (products of bf16 numbers are exact in float32: the bar is the float32 one), ragged shape,
  sums ADDED to non-zero gw / gb."""
  rng = np.random.RandomState(40 + cin + table)
  nseg, Bseg, H, W, cout = (3, 2, 45, 70, 8) if table else (1, 3, 77, 150, 8)
  x32 = [torch.tensor(rng.randn(Bseg, H, W, cin).astype(np.float32), device=cuda) for _ in range(nseg)]
  xb = [t.to(torch.bfloat16) for t in x32]
  du = [torch.tensor(rng.randn(Bseg, H, W, cout).astype(np.float32), device=cuda).to(torch.bfloat16) for _ in range(nseg)]
  xs = xb if cin == 8 else x32        # 8 -> 8: the stored bf16 activation; 4 -> 8: the float32 image
  fmt = (1 if cin == 8 else 0) | 2    # bit 0: x stored as bf16, bit 1: dU stored as bf16
  lib = rn.lib()
  nws = lib.ra_conv3x3_wgrad_workspace_floats(cin, cout, nseg * Bseg, H, W)
  ws = torch.full((nws,), float('nan'), device=cuda)
  g0w, g0b = rng.randn(3, 3, cin, cout).astype(np.float32), rng.randn(cout).astype(np.float32)
  gw, gb = torch.tensor(g0w, device=cuda), torch.tensor(g0b, device=cuda)
  if table:
    tab = torch.zeros(128, dtype=torch.int64, device=cuda)
    for off, ts in ((0, xs), (64, du)):
      host = (C.c_void_p * nseg)(*[t.data_ptr() for t in ts])
      rn.check(lib.ra_ptr_table(host, nseg, tab.data_ptr() + 8 * off, rn.stream_ptr()), 'ptr_table')
    rn.check(lib.ra_conv3x3_wgrad_multi_acc_f32(tab.data_ptr(), tab.data_ptr() + 8 * 64, nseg, cin, Bseg, H, W, 0, cout, rn.ptr(ws), nws,
                                                None, cin, 0, rn.ptr(gw), rn.ptr(gb), 1 | (fmt << 1), rn.stream_ptr()), 'wgrad_multi')
  else:
    rn.check(lib.ra_conv3x3_wgrad_acc_bf16_f32(rn.ptr(xs[0]), cin, Bseg, H, W, 0, rn.ptr(du[0]), cout, rn.ptr(ws), nws, None, cin, 0,
                                               rn.ptr(gw), rn.ptr(gb), fmt, rn.stream_ptr()), 'wgrad_bf16')
  rw, rb = _wgrad_ref(torch.cat([t.float() for t in xb]), torch.cat([t.float() for t in du]))
  rw, rb = rw + torch.tensor(g0w, device=cuda).double(), rb + torch.tensor(g0b, device=cuda).double()
  assert _rel(gw, rw) < 2e-5 and _rel(gb, rb) < 2e-5, (_rel(gw, rw), _rel(gb, rb))
