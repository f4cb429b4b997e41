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
