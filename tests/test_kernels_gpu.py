"""Per-kernel parity: every HIP kernel (through the C ABI) against the NumPy oracle on the same
seeded inputs.  Tolerances are float32 round-off class (the north star's bar is 1e-3 on masks)."""
import numpy as np
import pytest
import torch

import ra_oracle as ora
import ra_ops as ops
import ra_native as rn

pytestmark = pytest.mark.gpu


def dev(a, cuda):
  return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda)


def relerr(a, b):
  return np.abs(a - b).max() / max(1e-6, np.abs(b).max())


CONV_CASES = [
    # B, H, W, Cin, Cout, pool, relu
    (2, 32, 32, 4, 8, 1, True),
    (2, 32, 32, 8, 8, 2, True),
    (1, 64, 48, 8, 16, 1, True),
    (1, 16, 16, 16, 16, 2, False),
    (3, 48, 48, 16, 32, 1, True),
    (2, 24, 24, 32, 32, 2, True),
    (2, 16, 16, 32, 64, 2, True),
    (2, 8, 8, 64, 64, 2, True),
    (1, 128, 128, 4, 8, 1, True),
    (8, 64, 64, 16, 16, 2, True),   # many workgroups -> big tile geometry
    (1, 12, 12, 64, 96, 2, True),   # KITTI attn cnn last layer (Cout 96 -> 128 padded)
    (1, 20, 36, 24, 16, 2, True),   # Cityscapes packed input (24 ch), ragged tiles
    (1, 6, 10, 4, 1, 1, True),      # Cout = 1, tiny
]


@pytest.mark.parametrize('B,H,W,Ci,Co,pool,relu', CONV_CASES)
def test_conv3x3(cuda, B, H, W, Ci, Co, pool, relu):
  rng = np.random.RandomState(B * 1000 + H + Ci + Co)
  x = rng.randn(B, H, W, Ci).astype(np.float32)
  w = (rng.randn(3, 3, Ci, Co) / np.sqrt(9 * Ci)).astype(np.float32)
  b = rng.randn(Co).astype(np.float32) * 0.1
  bn = (rng.randn(Co) * 0.2, rng.uniform(0.5, 1.5, Co) * rng.choice([-1, 1], Co),
        rng.randn(Co) * 0.2, rng.uniform(0.5, 1.5, Co))
  bn = tuple(a.astype(np.float32) for a in bn)
  ref = ora.conv2d(x.astype(np.float64), w.astype(np.float64)) + b
  ref = ora.batch_norm_eval(ref, *[a.astype(np.float64) for a in bn])
  if relu:
    ref = ora.relu(ref)
  if pool > 1:
    ref = ora.max_pool(ref, pool)
  wp = dev(ops.pack_conv_weights(w), cuda)
  sc, sh = ops.fold_bn(b, Co, bn)
  y = ops.conv3x3(dev(x, cuda), wp, dev(sc, cuda), dev(sh, cuda), Co, relu=relu, pool=pool)
  torch.cuda.synchronize()
  y = y.cpu().numpy()
  assert y.shape == ref.shape
  assert relerr(y, ref) < 2e-5


def _bf16(a):
  """float32 -> nearest bf16 (ties to even) -> float32: what v_cvt_pk_bf16_f32 does to an operand."""
  return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).bfloat16().float().numpy()


@pytest.mark.parametrize('B,H,W,Ci,Co,relu', [
    (2, 32, 32, 4, 8, True), (1, 64, 48, 8, 16, True), (3, 48, 48, 16, 32, True), (2, 24, 24, 32, 32, False),
    (2, 16, 16, 32, 64, True), (1, 12, 12, 64, 96, True), (1, 20, 36, 24, 16, True), (8, 64, 64, 16, 16, True),
    (2, 40, 56, 12, 8, True)])
def test_conv3x3_bf16_operands(cuda, B, H, W, Ci, Co, relu):
  """ra_conv3x3_bf16ops_f32 (the training step's compute_dtype = 'bf16'): bf16 operands, float32 accumulation.
  Against the float64 oracle ON THE ROUNDED OPERANDS the bar is float32 round-off (products of two bf16 numbers are
  exact in float32, only the summation order differs) — that pins the rounding mode and the k-packing; against the
  oracle on the unrounded operands the bar is the bf16 one: 2^-8 per operand over a 9*Cin-term sum, 1e-2 of the
  output scale."""
  rng = np.random.RandomState(B * 1000 + H + Ci + Co)
  x = rng.randn(B, H, W, Ci).astype(np.float32)
  w = (rng.randn(3, 3, Ci, Co) / np.sqrt(9 * Ci)).astype(np.float32)
  b = (rng.randn(Co) * 0.1).astype(np.float32)

  def ref_of(xa, wa):
    r = ora.conv2d(xa.astype(np.float64), wa.astype(np.float64)) + b
    return ora.relu(r) if relu else r
  wp = dev(ops.pack_conv_weights(w), cuda)
  sc, sh = ops.fold_bn(b, Co, None)
  y = ops.conv3x3(dev(x, cuda), wp, dev(sc, cuda), dev(sh, cuda), Co, relu=relu, pool=1, bf16=True)
  y32 = ops.conv3x3(dev(x, cuda), wp, dev(sc, cuda), dev(sh, cuda), Co, relu=relu, pool=1)
  torch.cuda.synchronize()
  y, y32 = y.cpu().numpy(), y32.cpu().numpy()
  assert relerr(y, ref_of(_bf16(x), _bf16(w))) < 2e-5
  assert relerr(y, ref_of(x, w)) < 1e-2
  assert np.abs(y - y32).max() > 0  # the bf16 kernel really ran (the float32 one is exact to 2e-5 of the unrounded oracle)


@pytest.mark.parametrize('B,H,W,Ci,Co,ups', [(2, 32, 32, 4, 8, 0), (1, 24, 40, 8, 16, 0), (2, 16, 48, 16, 32, 0),
                                            (1, 16, 16, 32, 64, 0), (2, 8, 12, 16, 8, 1), (1, 33, 35, 12, 16, 0),
                                            (1, 16, 16, 64, 96, 0)])
def test_conv3x3_wgrad_bf16_operands(cuda, B, H, W, Ci, Co, ups):
  """ra_conv3x3_wgrad_bf16ops_f32: dW[ky,kx,ci,co] = sum over pixels of x[.. + tap, ci] * du[.., co] and db = sum du,
  with x and du rounded to bf16 (float32 sums).  Same two bars as the forward kernel."""
  rng = np.random.RandomState(7 * B + H + Ci + Co)
  Hs, Ws = (H // 2, W // 2) if ups else (H, W)
  x = rng.randn(B, Hs, Ws, Ci).astype(np.float32)
  du = rng.randn(B, H, W, Co).astype(np.float32)

  def ref_of(xa, da):
    xa, da = xa.astype(np.float64), da.astype(np.float64)
    if ups:  # the conv that ran saw x zero-stuffed at the odd positions (nnlib.dcnn's stride-2 transposed conv)
      z = np.zeros((B, H, W, Ci))
      z[:, 1::2, 1::2] = xa
      xa = z
    xp = np.pad(xa, ((0, 0), (1, 1), (1, 1), (0, 0)))
    dw = np.zeros((3, 3, Ci, Co))
    for ky in range(3):
      for kx in range(3):
        dw[ky, kx] = np.einsum('bhwc,bhwd->cd', xp[:, ky:ky + H, kx:kx + W], da)
    return dw, da.sum(axis=(0, 1, 2))
  nws = rn.lib().ra_conv3x3_wgrad_workspace_floats(Ci, Co, B, H, W)
  ws = torch.empty(nws, device=cuda)
  out = {}
  xd, dud = dev(x, cuda), dev(du, cuda)
  for name, fn in (('bf16', rn.lib().ra_conv3x3_wgrad_bf16ops_f32), ('f32', rn.lib().ra_conv3x3_wgrad_f32)):
    dw, db = torch.empty(3, 3, Ci, Co, device=cuda), torch.empty(Co, device=cuda)
    ops.check(fn(ops.ptr(xd), Ci, B, Hs, Ws, ups, ops.ptr(dud), Co, ops.ptr(ws), nws, ops.ptr(dw), ops.ptr(db),
                 rn.stream_ptr()), name)
    torch.cuda.synchronize()
    out[name] = (dw.cpu().numpy(), db.cpu().numpy())
  rw, rb = ref_of(_bf16(x), _bf16(du))
  assert relerr(out['bf16'][0], rw) < 2e-5 and relerr(out['bf16'][1], rb) < 2e-5
  uw, ub = ref_of(x, du)
  assert relerr(out['bf16'][0], uw) < 1e-2 and relerr(out['bf16'][1], ub) < 1e-2
  assert relerr(out['f32'][0], uw) < 2e-5
  assert np.abs(out['bf16'][0] - out['f32'][0]).max() > 0


@pytest.mark.parametrize('B,H,W,Cx,Cs,Co,stride', [
    (2, 6, 6, 32, 0, 32, 2),
    (2, 12, 12, 32, 0, 32, 1),
    (1, 12, 12, 16, 16, 8, 2),
    (2, 24, 24, 16, 16, 16, 1),
    (1, 48, 48, 8, 0, 1, 1),
    (1, 6, 6, 96, 0, 64, 2),
    (1, 12, 12, 64, 64, 64, 1),
    (1, 48, 48, 16, 13, 1, 1),  # skip with 13 real channels padded to 16
])
def test_conv_transpose(cuda, B, H, W, Cx, Cs, Co, stride):
  rng = np.random.RandomState(H * 7 + Cx + Cs + Co + stride)
  x = rng.randn(B, H, W, Cx).astype(np.float32)
  s = rng.randn(B, H, W, Cs).astype(np.float32) if Cs else None
  Cin = Cx + Cs
  w = (rng.randn(3, 3, Co, Cin) / np.sqrt(9 * Cin)).astype(np.float32)
  b = rng.randn(Co).astype(np.float32) * 0.1
  xin = x if s is None else np.concatenate([x, s], axis=3)
  ref = ora.relu(ora.conv2d_transpose(xin.astype(np.float64), w.astype(np.float64), stride) + b)
  cs_k = -(-Cs // 4) * 4
  cmap = list(range(Cx)) + [Cx + k if k < Cs else -1 for k in range(cs_k)]
  wp = dev(ops.pack_conv_weights(w, cin_kernel=Cx + cs_k, chan_map=cmap, transposed=True), cuda)
  sc, sh = ops.fold_bn(b, Co, None)
  sk = None
  if s is not None:
    sk = dev(np.pad(s, ((0, 0), (0, 0), (0, 0), (0, cs_k - Cs))), cuda)
  y = ops.conv3x3(dev(x, cuda), wp, dev(sc, cuda), dev(sh, cuda), Co, relu=True, pool=1, src1=sk,
                  upsample=(stride == 2))
  torch.cuda.synchronize()
  y = y.cpu().numpy()
  assert y.shape == ref.shape
  assert relerr(y, ref) < 2e-5


PAIR_CASES = [
    # B, H, W, Cin, CoutA, CoutB, poolB, upsA
    (2, 64, 64, 4, 8, 8, 2, False),     # ctrl L0+L1
    (1, 48, 80, 8, 16, 16, 2, False),   # ctrl L2+L3
    (2, 32, 32, 16, 32, 32, 2, False),  # ctrl L4+L5
    (3, 48, 48, 4, 8, 8, 2, False),     # attn a0+a1 (narrow tiles)
    (2, 24, 24, 8, 16, 16, 2, False),   # a2+a3
    (2, 12, 12, 16, 32, 32, 2, False),  # a4+a5
    (2, 6, 6, 32, 32, 32, 1, True),     # dcnn d0 (stride-2 transposed) + d1
    (2, 12, 12, 32, 16, 16, 1, True),   # d2+d3
    (1, 24, 24, 16, 8, 8, 1, True),     # d4+d5
    (1, 20, 36, 8, 8, 1, 1, False),     # CoutB = 1, ragged tiles
    (8, 128, 128, 4, 8, 8, 2, False),   # many tiles -> big geometry
    (2, 50, 70, 4, 8, 8, 2, False),     # N-packed kernel, ragged tiles on both edges
    (1, 18, 34, 8, 8, 5, 2, False),     # N-packed kernel, Cin = 8, CoutB < 8
    (1, 16, 32, 4, 8, 8, 2, False),     # N-packed kernel, exactly one tile
]


@pytest.mark.parametrize('B,H,W,Ci,Ca,Cb,poolB,ups', PAIR_CASES)
def test_conv_pair(cuda, B, H, W, Ci, Ca, Cb, poolB, ups):
  assert ops.conv_pair_supported(Ci, Ca, Cb)
  rng = np.random.RandomState(H * 3 + Ci + Ca + Cb)
  x = rng.randn(B, H, W, Ci).astype(np.float32)
  mk_bn = lambda c: tuple(a.astype(np.float32) for a in (
      rng.randn(c) * 0.2, rng.uniform(0.5, 1.5, c) * rng.choice([-1, 1], c), rng.randn(c) * 0.2,
      rng.uniform(0.5, 1.5, c)))
  bA, bB, bnA, bnB = rng.randn(Ca).astype(np.float32) * 0.1, rng.randn(Cb).astype(np.float32) * 0.1, \
      mk_bn(Ca), mk_bn(Cb)
  f64 = lambda t: tuple(a.astype(np.float64) for a in t)
  if ups:
    wA = (rng.randn(3, 3, Ca, Ci) / np.sqrt(9 * Ci)).astype(np.float32)
    wB = (rng.randn(3, 3, Cb, Ca) / np.sqrt(9 * Ca)).astype(np.float32)
    h = ora.conv2d_transpose(x.astype(np.float64), wA.astype(np.float64), 2) + bA
    h = ora.relu(ora.batch_norm_eval(h, *f64(bnA)))
    ref = ora.conv2d_transpose(h, wB.astype(np.float64), 1) + bB
  else:
    wA = (rng.randn(3, 3, Ci, Ca) / np.sqrt(9 * Ci)).astype(np.float32)
    wB = (rng.randn(3, 3, Ca, Cb) / np.sqrt(9 * Ca)).astype(np.float32)
    h = ora.relu(ora.batch_norm_eval(ora.conv2d(x.astype(np.float64), wA.astype(np.float64)) + bA,
                                     *f64(bnA)))
    ref = ora.conv2d(h, wB.astype(np.float64)) + bB
  ref = ora.relu(ora.batch_norm_eval(ref, *f64(bnB)))
  if poolB > 1:
    ref = ora.max_pool(ref, poolB)
  wpA = dev(ops.pack_conv_weights(wA, transposed=ups), cuda)
  wpB = dev(ops.pack_conv_weights(wB, transposed=ups), cuda)
  scA, shA = ops.fold_bn(bA, Ca, bnA)
  scB, shB = ops.fold_bn(bB, Cb, bnB)
  y = ops.conv_pair(dev(x, cuda), wpA, dev(scA, cuda), dev(shA, cuda), Ca, wpB, dev(scB, cuda),
                    dev(shB, cuda), Cb, poolB=poolB, upsampleA=ups)
  torch.cuda.synchronize()
  y = y.cpu().numpy()
  assert y.shape == ref.shape
  assert relerr(y, ref) < 3e-5


def _ctrl_setup(opt, seed):
  d = ora.derive(opt)
  P = ora.random_params(opt, seed)
  return d, P


@pytest.mark.parametrize('arch,H,W,flags', [
    ('cvppp', 128, 128, {}),
    ('cvppp', 224, 224, {'squash_ctrl_params': True}),   # G = 49: not a multiple of 4
    ('kitti', 128, 448, {}),
    ('cvppp', 512, 512, {'fixed_var': True, 'num_ctrl_mlp_layers': 2, 'num_glimpse_mlp_layers': 3}),
])
def test_controller(cuda, arch, H, W, flags):
  opt = ora.make_opt(arch, H, W, 2, **flags)
  d, P = _ctrl_setup(opt, 3)
  B = 3
  rng = np.random.RandomState(5)
  Cf = d['ccnn_channels'][-1]
  feat = np.maximum(rng.randn(B, d['G'], Cf), 0).astype(np.float32)
  P64 = {k: v.astype(np.float64) for k, v in P.items()}
  h, co, gm = ora._controller(d, P64, feat.astype(np.float64), np.dtype(np.float64))
  cn, ls, ctr, size, lv = ora._decode_ctrl(d, co, np.dtype(np.float64))
  desc = ops.make_ctrl_desc(d['G'], Cf, d['hid'], d['iters'], d['n_gmlp'], d['n_cmlp'],
                            opt['ctrl_mlp_dim'], H, W, 48, 48, d['squash'], d['fixed_var'],
                            d['dynamic_var'], d['fixed_gamma'])
  lstm = {k[len('ctrl_lstm_'):]: v for k, v in P.items() if k.startswith('ctrl_lstm_')}
  gmw = [(P['glimpse_mlp_w_%d' % i], P['glimpse_mlp_b_%d' % i]) for i in range(d['n_gmlp'])]
  cmw = [(P['ctrl_mlp_w_%d' % i], P['ctrl_mlp_b_%d' % i]) for i in range(d['n_cmlp'])]
  wp = dev(ops.pack_ctrl_weights(desc, lstm, gmw, cmw), cuda)
  z = lambda *s: torch.zeros(s, dtype=torch.float32, device=cuda)
  h_last, ctrl_out, gmaps, attn = z(B, d['hid']), z(B, 9), z(B, d['iters'], d['G']), z(B, 16)
  ops.controller(desc, dev(feat, cuda), wp, h_last, ctrl_out, gmaps, attn)
  torch.cuda.synchronize()
  assert relerr(h_last.cpu().numpy(), h) < 5e-5
  assert np.abs(ctrl_out.cpu().numpy() - co).max() < 5e-5
  assert np.abs(gmaps.cpu().numpy() - gm).max() < 1e-5
  a = attn.cpu().numpy()
  assert np.abs(a[:, 0:2] - ctr).max() < 1e-3 * max(H, W) / 100
  assert relerr(a[:, 2:4], size) < 1e-4
  assert np.abs(a[:, 4:6] - lv).max() < 1e-4
  assert np.abs(a[:, 9:11] - cn).max() < 1e-4 and np.abs(a[:, 11:13] - ls).max() < 1e-4
  if d['fixed_gamma']:
    assert (a[:, 6] == 1.0).all() and (a[:, 8] == 2.0).all()
  else:
    assert relerr(a[:, 6], np.exp(co[:, 6])) < 1e-4 and np.abs(a[:, 8] - co[:, 8]).max() < 1e-4
  assert relerr(a[:, 7], np.exp(co[:, 7])) < 1e-4


@pytest.mark.parametrize('arch,H,W,flags,B', [
    ('cvppp', 128, 128, {}, 5),
    ('cvppp', 224, 224, {'squash_ctrl_params': True}, 5),   # G = 49 -> gs = 4, padded logits
    ('kitti', 128, 448, {}, 5),
    ('cvppp', 512, 512, {'fixed_var': True, 'num_ctrl_mlp_layers': 2, 'num_glimpse_mlp_layers': 3}, 5),
    ('cvppp', 512, 512, {'num_glimpse_mlp_layers': 1}, 5),
    ('cvppp', 512, 512, {}, 8),    # the XCD-local form: one team of 16 workgroups on every XCD
    ('cvppp', 512, 512, {}, 11),   # 9-14 images: the grid (16, B) form by default (RA_CTRL_XCD=2: two teams on three XCDs, one on five)
    ('cvppp', 128, 128, {}, 14),   # ... the residency limit: 224 workgroups of the 256 launched take a role
    ('cvppp', 128, 128, {}, 1),
])
def test_controller_split(cuda, arch, H, W, flags, B):
  """The 16-workgroup LDS-stationary controller: same maths, exchanged through tagged granules;
  launched three times on the same workspace (generation tags, as under HIP-graph replay; the XCD-local form's role
  tickets only ever count up)."""
  opt = ora.make_opt(arch, H, W, 2, **flags)
  d, P = _ctrl_setup(opt, 4)
  Cf = d['ccnn_channels'][-1]
  desc = ops.make_ctrl_desc(d['G'], Cf, d['hid'], d['iters'], d['n_gmlp'], d['n_cmlp'],
                            opt['ctrl_mlp_dim'], H, W, 48, 48, d['squash'], d['fixed_var'],
                            d['dynamic_var'], d['fixed_gamma'])
  assert ops.ctrl_split_supported(desc)
  lstm = {k[len('ctrl_lstm_'):]: v for k, v in P.items() if k.startswith('ctrl_lstm_')}
  gmw = [(P['glimpse_mlp_w_%d' % i], P['glimpse_mlp_b_%d' % i]) for i in range(d['n_gmlp'])]
  cmw = [(P['ctrl_mlp_w_%d' % i], P['ctrl_mlp_b_%d' % i]) for i in range(d['n_cmlp'])]
  wp = dev(ops.pack_ctrl_split_weights(desc, lstm, gmw, cmw), cuda)
  ws, status = ops.ctrl_split_workspace(desc, B, cuda)
  P64 = {k: v.astype(np.float64) for k, v in P.items()}
  z = lambda *s: torch.full(s, 7.0, dtype=torch.float32, device=cuda)
  for rep in range(3):
    rng = np.random.RandomState(50 + rep)
    feat = np.maximum(rng.randn(B, d['G'], Cf), 0).astype(np.float32)
    h, co, gm = ora._controller(d, P64, feat.astype(np.float64), np.dtype(np.float64))
    cn, ls, ctr, size, lv = ora._decode_ctrl(d, co, np.dtype(np.float64))
    h_last, ctrl_out, gmaps, attn = z(B, d['hid']), z(B, 9), z(B, d['iters'], d['G']), z(B, 16)
    ops.controller_split(desc, dev(feat, cuda), wp, h_last, ctrl_out, gmaps, attn, ws, status)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    assert relerr(h_last.cpu().numpy(), h) < 5e-5
    assert np.abs(ctrl_out.cpu().numpy() - co).max() < 5e-5
    assert np.abs(gmaps.cpu().numpy() - gm).max() < 1e-5
    a = attn.cpu().numpy()
    assert np.abs(a[:, 0:2] - ctr).max() < 1e-3 * max(H, W) / 100
    assert relerr(a[:, 2:4], size) < 1e-4 and np.abs(a[:, 4:6] - lv).max() < 1e-4


@pytest.mark.parametrize('arch,H,W,flags,B', [
    ('cvppp', 128, 128, {}, 5),
    ('cvppp', 224, 224, {'squash_ctrl_params': True}, 8),   # G = 49 -> gs = 4, padded logits
    ('kitti', 128, 448, {}, 16),                             # full groups only
    ('cvppp', 512, 512, {'fixed_var': True, 'num_ctrl_mlp_layers': 2}, 11),  # full groups + a ragged one
    ('cvppp', 512, 512, {'num_glimpse_mlp_layers': 1}, 1),
    ('cvppp', 512, 512, {}, 16),       # the headline's launch: a pipeline slot of two batches of 8 (bench.py, coalesce = 2)
    ('cityscapes', 256, 512, {}, 16),  # cfg5's slot: four batches of 4
    ('kitti', 128, 448, {}, 32),       # cfg3's slot: two batches of 16
])
def test_controller_batch(cuda, arch, H, W, flags, B):
  """K2b, the split controller with its weight slices shared by groups of 4 images (8 for launches of more than 8): the oracle's recurrence,
  three launches on one workspace (generation tags), full and ragged groups, and agreement with the
  per-image split form."""
  opt = ora.make_opt(arch, H, W, 2, **flags)
  d, P = _ctrl_setup(opt, 4)
  Cf = d['ccnn_channels'][-1]
  desc = ops.make_ctrl_desc(d['G'], Cf, d['hid'], d['iters'], d['n_gmlp'], d['n_cmlp'],
                            opt['ctrl_mlp_dim'], H, W, 48, 48, d['squash'], d['fixed_var'],
                            d['dynamic_var'], d['fixed_gamma'])
  assert ops.ctrl_batch_supported(desc)
  assert ops.ctrl_batch_group(desc, B) == (8 if B > 8 else 4)
  # a third glimpse-MLP layer makes the slice + a group of 8's vectors exceed 160 KB of LDS: such launches keep groups of 4
  big = ops.make_ctrl_desc(d['G'], Cf, d['hid'], d['iters'], 3, d['n_cmlp'], opt['ctrl_mlp_dim'], H, W, 48, 48,
                           d['squash'], d['fixed_var'], d['dynamic_var'], d['fixed_gamma'])
  if d['G'] * 4 >= 1024:  # (with few logits per slice the larger form still fits)
    assert ops.ctrl_split_supported(big) and ops.ctrl_batch_supported(big) and ops.ctrl_batch_group(big, 16) == 4
  lstm = {k[len('ctrl_lstm_'):]: v for k, v in P.items() if k.startswith('ctrl_lstm_')}
  gmw = [(P['glimpse_mlp_w_%d' % i], P['glimpse_mlp_b_%d' % i]) for i in range(d['n_gmlp'])]
  cmw = [(P['ctrl_mlp_w_%d' % i], P['ctrl_mlp_b_%d' % i]) for i in range(d['n_cmlp'])]
  wp = dev(ops.pack_ctrl_split_weights(desc, lstm, gmw, cmw), cuda)
  ws, status = ops.ctrl_batch_workspace(desc, B, cuda)
  P64 = {k: v.astype(np.float64) for k, v in P.items()}
  z = lambda *s: torch.full(s, 7.0, dtype=torch.float32, device=cuda)
  for rep in range(3):
    rng = np.random.RandomState(70 + rep)
    feat = np.maximum(rng.randn(B, d['G'], Cf), 0).astype(np.float32)
    h, co, gm = ora._controller(d, P64, feat.astype(np.float64), np.dtype(np.float64))
    cn, ls, ctr, size, lv = ora._decode_ctrl(d, co, np.dtype(np.float64))
    h_last, ctrl_out, gmaps, attn = z(B, d['hid']), z(B, 9), z(B, d['iters'], d['G']), z(B, 16)
    # launch 0: agent-scope exchange; launches 1, 2: the XCD-local form (round 6), group g on XCD (g + offset) % 8 — the same
    # workspace (generation tags and role tickets only ever count up)
    ops.controller_batch(desc, dev(feat, cuda), wp, h_last, ctrl_out, gmaps, attn, ws, status, xcd_offset=(-1, 0, 5)[rep])
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    assert relerr(h_last.cpu().numpy(), h) < 5e-5
    assert np.abs(ctrl_out.cpu().numpy() - co).max() < 5e-5
    assert np.abs(gmaps.cpu().numpy() - gm).max() < 1e-5
    a = attn.cpu().numpy()
    assert np.abs(a[:, 0:2] - ctr).max() < 1e-3 * max(H, W) / 100
    assert relerr(a[:, 2:4], size) < 1e-4 and np.abs(a[:, 4:6] - lv).max() < 1e-4
  if B <= 14:
    ws2, st2 = ops.ctrl_split_workspace(desc, B, cuda)
    h2, c2, g2, a2 = z(B, d['hid']), z(B, 9), z(B, d['iters'], d['G']), z(B, 16)
    ops.controller_split(desc, dev(feat, cuda), wp, h2, c2, g2, a2, ws2, st2)
    torch.cuda.synchronize()
    assert np.abs(c2.cpu().numpy() - ctrl_out.cpu().numpy()).max() < 2e-5


def _attn_rec(B, H, W, rng, big_var=False):
  rec = np.zeros((B, 16), np.float32)
  rec[:, 0] = rng.uniform(0.1, 0.9, B) * H
  rec[:, 1] = rng.uniform(0.1, 0.9, B) * W
  rec[:, 2] = rng.uniform(0.15, 0.7, B) * H
  rec[:, 3] = rng.uniform(0.15, 0.7, B) * W
  rec[:, 4] = np.log(rec[:, 2] / 48.0)
  rec[:, 5] = np.log(rec[:, 3] / 48.0)
  if big_var:
    rec[:, 4:6] += 4.0  # very wide taps: bands cover (almost) the whole image
  rec[:, 6] = rng.uniform(0.5, 2.0, B)
  rec[:, 7] = rng.uniform(0.5, 2.0, B)
  rec[:, 8] = rng.uniform(0.5, 2.5, B)
  return rec


@pytest.mark.parametrize('H,W,big', [(128, 128, False), (96, 160, True)])
def test_dense_attention_operators(cuda, H, W, big):
  """The literal operators of modellib — get_gaussian_filter (the [L,F] bank) and extract_patch for given banks —
  against the oracle (the decode loop and the training step use the banded on-the-fly kernels instead)."""
  rng = np.random.RandomState(H + W)
  B, C = 3, 8
  rec = _attn_rec(B, H, W, rng, big)
  rec[0, 0], rec[0, 1] = 0.02 * H, 0.97 * W  # box partly outside the image
  r64 = rec.astype(np.float64)
  fy_ref = ora.get_gaussian_filter(r64[:, 0], r64[:, 2], r64[:, 4], H, 48)
  fx_ref = ora.get_gaussian_filter(r64[:, 1], r64[:, 3], r64[:, 5], W, 48)
  fy = ops.gaussian_filter(dev(rec[:, 0], cuda), dev(rec[:, 2], cuda), dev(rec[:, 4], cuda), H, 48)
  fx = ops.gaussian_filter(dev(rec[:, 1], cuda), dev(rec[:, 3], cuda), dev(rec[:, 5], cuda), W, 48)
  assert relerr(fy.cpu().numpy(), fy_ref) < 1e-4 and relerr(fx.cpu().numpy(), fx_ref) < 1e-4
  img = rng.rand(B, H, W, C).astype(np.float32)
  ref = ora.extract_patch(img.astype(np.float64), fy_ref, fx_ref, C)
  pd = ops.extract_patch_dense(dev(img, cuda), fy, fx)
  assert np.abs(pd.cpu().numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize('B,H,W,Ci,Co,pool,relu', [(2, 32, 48, 16, 32, 1, True), (1, 64, 32, 32, 32, 2, True), (3, 16, 16, 32, 64, 2, True),
                                                    (2, 48, 32, 16, 64, 2, False), (8, 128, 128, 32, 32, 2, True), (2, 32, 32, 32, 16, 1, False),
                                                    (1, 16, 32, 16, 48, 2, True),
                                                    (8, 32, 32, 64, 64, 2, True), (2, 8, 48, 64, 64, 1, True), (3, 24, 16, 64, 32, 2, False),
                                                    (1, 40, 32, 64, 16, 1, True),  # Cin = 64: 8-row tiles, 16 couts per workgroup
                                                    # round 6: ragged last tiles (KITTI's 56- / 28-pixel maps, the 24- / 12-pixel patch maps), Cin = 24
                                                    (2, 16, 56, 32, 64, 1, True), (3, 16, 56, 64, 64, 2, True), (2, 8, 28, 64, 64, 2, False),
                                                    (2, 24, 24, 16, 32, 2, True), (3, 12, 12, 64, 96, 2, True), (2, 48, 48, 16, 16, 1, True),
                                                    (2, 64, 112, 16, 16, 2, True), (2, 32, 48, 24, 16, 2, True), (1, 20, 36, 24, 32, 1, False)])
def test_conv_split_precision(cuda, B, H, W, Ci, Co, pool, relu):
  """K1s (ra_conv_split_f32, round 5): the direct 3x3 layer on the bf16 matrix pipe — every operand the exact sum of three bf16
  pieces, six of the nine piece products — against the float64 oracle at the float32 kernels' bar (2e-5 of the output scale),
  with operands of mixed magnitude (the dropped piece products are below 2^-24 of a product whatever the exponents), and
  against K1 on the same inputs."""
  rng = np.random.RandomState(B * 100 + H + Ci + Co)
  x = (rng.randn(B, H, W, Ci) * np.exp(rng.uniform(-3, 3, (B, H, W, Ci)))).astype(np.float32)
  w = (rng.randn(3, 3, Ci, Co) / np.sqrt(9 * Ci)).astype(np.float32)
  b = (rng.randn(Co) * 0.1).astype(np.float32)
  bn = (rng.randn(Co).astype(np.float32) * 0.1, rng.uniform(0.5, 1.5, Co).astype(np.float32), rng.randn(Co).astype(np.float32) * 0.1,
        rng.uniform(0.5, 1.5, Co).astype(np.float32))
  sc, sh = ops.fold_bn(b, Co, bn)
  ref = (ora.conv2d(x.astype(np.float64), w.astype(np.float64))) * sc[:Co].astype(np.float64) + sh[:Co].astype(np.float64)
  if relu:
    ref = ora.relu(ref)
  if pool == 2:
    ref = ora.max_pool(ref, 2)
  assert ops.conv_split_supported(Ci, Co, pool, H, W) and not ops.conv_split_supported(8, Co, pool, H, W)
  wp = torch.from_numpy(ops.pack_split_weights(w)).to(cuda)
  y = ops.conv_split(dev(x, cuda), wp, dev(sc, cuda), dev(sh, cuda), Co, relu=relu, pool=pool)
  y1 = ops.conv3x3(dev(x, cuda), dev(ops.pack_conv_weights(w), cuda), dev(sc, cuda), dev(sh, cuda), Co, relu=relu, pool=pool)
  torch.cuda.synchronize()
  assert y.shape == ref.shape
  e_split, e_k1 = relerr(y.cpu().numpy(), ref), relerr(y1.cpu().numpy(), ref)
  assert e_split < 2e-5 and e_split < 4 * e_k1 + 1e-7, (e_split, e_k1)
  # the device packer (the training step's filters change every step) writes the same words as the host one ...
  wd = ops.pack_split_weights_dev(dev(w, cuda), Ci, Co)
  assert torch.equal(wd.view(torch.int32), wp.view(torch.int32))
  # ... and, transposed, the packing of the layer's DATA GRADIENT: dx = conv(du, flip(w) with in / out swapped) (nnlib.py:229-253
  # differentiated); K1s takes it when the layer's channel counts allow
  if ops.conv_split_supported(Co, Ci, 1, H, W):
    du = rng.randn(B, H, W, Co).astype(np.float32)
    wt = np.ascontiguousarray(w[::-1, ::-1].transpose(0, 1, 3, 2))  # [3,3,Co,Ci]: the equivalent direct filter
    ref_dx = ora.conv2d(du.astype(np.float64), wt.astype(np.float64))
    wpt = ops.pack_split_weights_dev(dev(w, cuda), Co, Ci, transposed=True)
    cpi = ops.cout_padded(Ci)
    dx = ops.conv_split(dev(du, cuda), wpt, torch.ones(cpi, device=cuda), torch.zeros(cpi, device=cuda), Ci, relu=False, pool=1)
    assert relerr(dx.cpu().numpy(), ref_dx) < 2e-5


@pytest.mark.parametrize('B,H,W,real,Ck,Co,pool', [(2, 64, 96, 13, 16, 16, 2), (2, 32, 64, 21, 24, 16, 2), (1, 128, 448, 13, 16, 16, 2),
                                                    (3, 16, 40, 21, 24, 32, 1), (2, 32, 32, 30, 32, 16, 2)])
def test_conv_split_first_layer_with_canvas_plane(cuda, B, H, W, real, Ck, Co, pool):
  """ra_conv_split_plane_f32 (round 6): K1s as the FIRST controller-CNN layer of the KITTI / Cityscapes architectures — the packed
  input concat(x, canvas, d_in, y_in) (full_model.py:640-661: 13 / 21 real channels in 16 / 24, the rest zero pad with garbage
  allowed) with the canvas in its own plane, the filter's rows found through the channel map — against the float64 oracle on the
  reference's concatenated input, and against K1 with the same plane."""
  rng = np.random.RandomState(B + H + W + real)
  D = 3
  img = rng.rand(B, H, W, Ck).astype(np.float32)
  img[..., D] = 777.0          # the canvas slot of the packed image is never read
  img[..., real:] = 555.0      # ... nor are the pad channels (their filter rows are zero)
  canvas = rng.rand(B, H, W).astype(np.float32)
  w = (rng.randn(3, 3, real, Co) / np.sqrt(9 * real)).astype(np.float32)
  b = (rng.randn(Co) * 0.1).astype(np.float32)
  bn = (rng.randn(Co).astype(np.float32) * 0.1, rng.uniform(0.5, 1.5, Co).astype(np.float32), rng.randn(Co).astype(np.float32) * 0.1,
        rng.uniform(0.5, 1.5, Co).astype(np.float32))
  sc, sh = ops.fold_bn(b, Co, bn)
  cmap = list(range(real)) + [-1] * (Ck - real)
  xin = img[..., :real].astype(np.float64).copy()
  xin[..., D] = canvas
  ref = ora.relu(ora.conv2d(xin, w.astype(np.float64)) * sc[:Co].astype(np.float64) + sh[:Co].astype(np.float64))
  if pool == 2:
    ref = ora.max_pool(ref, 2)
  assert ops.conv_split_supported(Ck, Co, pool, H, W)
  wp = torch.from_numpy(ops.pack_split_weights(w, cin_kernel=Ck, chan_map=cmap)).to(cuda)
  y = ops.conv_split(dev(img, cuda), wp, dev(sc, cuda), dev(sh, cuda), Co, relu=True, pool=pool, plane=dev(canvas, cuda), plane_chan=D)
  y1 = ops.conv3x3(dev(img, cuda), dev(ops.pack_conv_weights(w, cin_kernel=Ck, chan_map=cmap), cuda), dev(sc, cuda), dev(sh, cuda), Co,
                   relu=True, pool=pool, plane=dev(canvas, cuda), plane_chan=D)
  torch.cuda.synchronize()
  assert y.shape == ref.shape
  e_split, e_k1 = relerr(y.cpu().numpy(), ref), relerr(y1.cpu().numpy(), ref)
  assert e_split < 2e-5 and e_split < 4 * e_k1 + 1e-7, (e_split, e_k1)


@pytest.mark.parametrize('H,W,big,cout,F', [(128, 128, False, 8, 48), (96, 160, True, 8, 48), (512, 512, False, 8, 48),
                                             (64, 96, False, 16, 48), (128, 128, True, 12, 32)])
def test_extract_fused_with_first_attention_cnn_layer(cuda, H, W, big, cout, F):
  """ra_extract_conv0_f32 (round 5): the extract and layer 0 of the attention CNN (3x3 SAME, folded BN, ReLU, no pool) as ONE
  launch — workgroup (tap j, image) reduces the rows of taps j-1, j, j+1 once and evaluates the conv's output row j from the
  three patch rows.  Against the float64 oracle (extract_patch, then conv2d + affine + ReLU) and against the two separate
  product launches; canvas as a plane, a box partly outside the image, wide dynamic_var bands, Cout 8 / 12 / 16, F 48 / 32."""
  rng = np.random.RandomState(H + 3 * W + cout)
  B = 3
  rec = _attn_rec(B, H, W, rng, big)
  rec[0, 0], rec[0, 1] = 0.02 * H, 0.97 * W
  r64 = rec.astype(np.float64)
  fy_ref = ora.get_gaussian_filter(r64[:, 0], r64[:, 2], r64[:, 4], H, F)
  fx_ref = ora.get_gaussian_filter(r64[:, 1], r64[:, 3], r64[:, 5], W, F)
  img = rng.rand(B, H, W, 4).astype(np.float32)
  canvas0 = rng.uniform(0, 0.6, (B, H, W)).astype(np.float32)
  img_ref = img.copy()
  img_ref[..., 3] = canvas0
  w0 = (rng.randn(3, 3, 4, cout) * 0.3).astype(np.float32)
  w0[:, :, 2, :] = 0  # an input channel the model does not feed: a zero row of the packed-order filter
  cp = ops.cout_padded(cout)
  sc = np.ones(cp, np.float32)
  sh = np.zeros(cp, np.float32)
  sc[:cout], sh[:cout] = rng.uniform(0.5, 1.5, cout), rng.randn(cout) * 0.2
  xp_ref = r64[:, 6].reshape(-1, 1, 1, 1) * ora.extract_patch(img_ref.astype(np.float64), fy_ref, fx_ref, 4)
  y_ref = ora.relu(ora.conv2d(xp_ref, w0.astype(np.float64)) * sc[:cout] + sh[:cout])
  assert ops.extract_conv0_supported(4, F, F, cout, 1) and not ops.extract_conv0_supported(8, F, F, cout, 1)
  for plane in (True, False):
    dimg = dev(img if plane else img_ref, cuda)
    dcv = dev(canvas0, cuda) if plane else None
    patch = torch.full((B, F, F, 4), 7.0, dtype=torch.float32, device=cuda)
    y0 = torch.full((B, F, F, cout), 7.0, dtype=torch.float32, device=cuda)
    ops.extract_conv0(dimg, 0, dev(rec, cuda), F, F, True, patch, dev(w0, cuda), dev(sc, cuda), dev(sh, cuda), cout, True, y0,
                      canvas=dcv, canvas_chan=3)
    torch.cuda.synchronize()
    assert np.abs(patch.cpu().numpy() - xp_ref).max() < 3e-5 * max(1.0, np.abs(xp_ref).max())
    assert np.abs(y0.cpu().numpy() - y_ref).max() < 5e-5 * max(1.0, np.abs(y_ref).max()), np.abs(y0.cpu().numpy() - y_ref).max()
    # ... and the two launches it replaces
    p2 = torch.zeros_like(patch)
    ops.extract_direct(dimg, 0, dev(rec, cuda), F, F, 4, True, p2, canvas=dcv, canvas_chan=3)
    y2 = ops.conv3x3(p2, dev(ops.pack_conv_weights(w0), cuda), dev(sc, cuda), dev(sh, cuda), cout, relu=True, pool=1)
    assert np.abs(p2.cpu().numpy() - patch.cpu().numpy()).max() < 2e-5 * max(1.0, np.abs(xp_ref).max())
    assert np.abs(y2.cpu().numpy() - y0.cpu().numpy()).max() < 5e-5 * max(1.0, np.abs(y_ref).max())


@pytest.mark.parametrize('H,W,big', [(128, 128, False), (96, 160, True), (512, 512, False)])
def test_direct_extract_paste_box(cuda, H, W, big):
  """The table-free attention kernels (weights computed on the fly), with the canvas either as a
  channel of the packed image or as its own plane."""
  rng = np.random.RandomState(H * 2 + W)
  B, C = 3, 8
  rec = _attn_rec(B, H, W, rng, big)
  rec[0, 0], rec[0, 1] = 0.02 * H, 0.97 * W  # box partly outside the image
  r64 = rec.astype(np.float64)
  fy_ref = ora.get_gaussian_filter(r64[:, 0], r64[:, 2], r64[:, 4], H, 48)
  fx_ref = ora.get_gaussian_filter(r64[:, 1], r64[:, 3], r64[:, 5], W, 48)
  img = rng.rand(B, H, W, C).astype(np.float32)
  canvas0 = rng.uniform(0, 0.6, (B, H, W)).astype(np.float32)
  img_ref = img.copy()
  img_ref[..., 3] = canvas0
  ref = r64[:, 6].reshape(-1, 1, 1, 1) * ora.extract_patch(img_ref.astype(np.float64), fy_ref, fx_ref, C)
  tol = 3e-5 * max(1.0, np.abs(ref).max())
  for plane in (False, True):
    dimg = dev(img if plane else img_ref, cuda)  # with a plane, channel 3 of img is ignored
    dcv = dev(canvas0, cuda) if plane else None
    patch = torch.zeros((B, 48, 48, C), dtype=torch.float32, device=cuda)
    ops.extract_direct(dimg, 0, dev(rec, cuda), 48, 48, C, True, patch, canvas=dcv, canvas_chan=3)
    torch.cuda.synchronize()
    assert np.abs(patch.cpu().numpy() - ref).max() < tol
    P = rng.randn(B, 48, 48, 1).astype(np.float32)
    yy = ora.extract_patch(P.astype(np.float64), np.transpose(fy_ref, (0, 2, 1)),
                           np.transpose(fx_ref, (0, 2, 1)), 1)[..., 0]
    yy = ora.sigmoid(np.exp(r64[:, 8]).reshape(-1, 1, 1) * yy - 5.0)
    for overwrite in (True, False):
      dimg = dev(img if plane else img_ref, cuda)
      dcv = dev(canvas0, cuda) if plane else None
      y_out = torch.zeros((B, 2, H, W), dtype=torch.float32, device=cuda)
      ops.paste_direct(dev(P, cuda), 0, dev(rec, cuda), -5.0, overwrite,
                       y_out.data_ptr() + H * W * 4, 2 * H * W, H, W, canvas=dcv,
                       img=None if plane else dimg, canvas_chan=-1 if plane else 3)
      torch.cuda.synchronize()
      yo = yy * (1 - canvas0.astype(np.float64)) if overwrite else yy
      got = y_out.cpu().numpy()
      assert (got[:, 0] == 0).all() and np.abs(got[:, 1] - yo).max() < 2e-5
      cv = dcv.cpu().numpy() if plane else dimg.cpu().numpy()[..., 3]
      assert np.abs(cv - np.maximum(yo, canvas0)).max() < 2e-5
      if not plane:
        gi = dimg.cpu().numpy()
        assert (gi[..., [0, 1, 2, 4, 5, 6, 7]] == img_ref[..., [0, 1, 2, 4, 5, 6, 7]]).all()
  # window-only paste: y_out prefilled with sigmoid(beta), canvas already >= sigmoid(beta)
  y_dead = 1.0 / (1.0 + np.exp(5.0))
  cfl = np.maximum(canvas0, np.float32(y_dead))
  dcv = dev(cfl, cuda)
  y_out = torch.full((B, 2, H, W), y_dead, dtype=torch.float32, device=cuda)
  ops.paste_direct(dev(P, cuda), 0, dev(rec, cuda), -5.0, False, y_out.data_ptr() + H * W * 4,
                   2 * H * W, H, W, canvas=dcv,
                   flags=ops.PASTE_Y_PREFILLED | ops.PASTE_CANVAS_FLOORED)
  torch.cuda.synchronize()
  got = y_out.cpu().numpy()
  assert (got[:, 0] == np.float32(y_dead)).all() and np.abs(got[:, 1] - yy).max() < 2e-5
  assert np.abs(dcv.cpu().numpy() - np.maximum(yy, cfl)).max() < 2e-5
  box = torch.zeros((B, H, W), dtype=torch.float32, device=cuda)
  ops.attn_box_direct(dev(rec, cuda), H, W, 48, 48, -5.0, box, H * W)
  ones = np.ones((B, 48, 48, 1))
  bref = ora.sigmoid(ora.extract_patch(ones * r64[:, 7].reshape(-1, 1, 1, 1),
                                       np.transpose(fy_ref, (0, 2, 1)),
                                       np.transpose(fx_ref, (0, 2, 1)), 1) - 5.0)
  torch.cuda.synchronize()
  assert np.abs(box.cpu().numpy() - bref[..., 0]).max() < 3e-5


def test_conv_with_canvas_plane(cuda):
  """Input channel 3 supplied by a separate [B,H,W] plane (single and fused-pair kernels)."""
  rng = np.random.RandomState(12)
  B, H, W = 2, 40, 56
  x = rng.randn(B, H, W, 4).astype(np.float32)
  plane = rng.randn(B, H, W).astype(np.float32)
  xr = x.copy()
  xr[..., 3] = plane
  wA = (rng.randn(3, 3, 4, 8) / 6).astype(np.float32)
  wB = (rng.randn(3, 3, 8, 8) / 8).astype(np.float32)
  hA = ora.relu(ora.conv2d(xr.astype(np.float64), wA.astype(np.float64)))
  refB = ora.max_pool(ora.relu(ora.conv2d(hA, wB.astype(np.float64))), 2)
  wpA, wpB = dev(ops.pack_conv_weights(wA), cuda), dev(ops.pack_conv_weights(wB), cuda)
  sc, sh = [dev(a, cuda) for a in ops.fold_bn(None, 8)]
  y = ops.conv3x3(dev(x, cuda), wpA, sc, sh, 8, relu=True, pool=1, plane=dev(plane, cuda), plane_chan=3)
  assert relerr(y.cpu().numpy(), hA) < 2e-5
  y2 = ops.conv_pair(dev(x, cuda), wpA, sc, sh, 8, wpB, sc, sh, 8, poolB=2, plane=dev(plane, cuda),
                     plane_chan=3)
  assert relerr(y2.cpu().numpy(), refB) < 3e-5


def test_dense_pool_affine_pack(cuda):
  rng = np.random.RandomState(0)
  B = 5
  x0, x1 = rng.randn(B, 256).astype(np.float32), rng.randn(B, 1152).astype(np.float32)
  W = (rng.randn(1408, 3) / 30).astype(np.float32)
  b = rng.randn(3).astype(np.float32)
  z = np.concatenate([x0, x1], 1).astype(np.float64) @ W + b
  for act, f in (('sigmoid', ora.sigmoid), ('softmax', ora.softmax), (None, lambda v: v),
                 ('relu', ora.relu), ('tanh', np.tanh)):
    out = torch.zeros((B, 3), dtype=torch.float32, device=cuda)
    ops.dense(dev(x0, cuda), dev(W, cuda), dev(b, cuda), act, out, 3, x1=dev(x1, cuda))
    torch.cuda.synchronize()
    assert np.abs(out.cpu().numpy() - f(z)).max() < 2e-5
  x = rng.randn(2, 7, 9, 5).astype(np.float32)
  for r in (2, 3):
    y = ops.max_pool(dev(x, cuda), r).cpu().numpy()
    assert (y == ora.max_pool(x, r)).all()
  sc, sh = rng.randn(5).astype(np.float32), rng.randn(5).astype(np.float32)
  y = ops.affine_act(dev(x, cuda), dev(sc, cuda), dev(sh, cuda), relu=True).cpu().numpy()
  assert np.abs(y - np.maximum(x * sc + sh, 0)).max() < 1e-6
  xi = rng.rand(2, 8, 8, 3).astype(np.float32)
  di, yi = rng.rand(2, 8, 8, 8).astype(np.float32), rng.rand(2, 8, 8, 1).astype(np.float32)
  packed = torch.full((2, 8, 8, 16), 7.0, dtype=torch.float32, device=cuda)
  ops.pack_input(dev(xi, cuda), dev(di, cuda), dev(yi, cuda), 16, packed)
  ref = np.concatenate([xi, np.zeros((2, 8, 8, 1), np.float32), di, yi,
                        np.zeros((2, 8, 8, 3), np.float32)], 3)
  assert (packed.cpu().numpy() == ref).all()


@pytest.mark.parametrize('shape', [(2, 64, 96), (1, 128, 160), (3, 48, 80)])
def test_first_layer_cache_vs_plain_pair(cuda, shape):
  """The cached form of the first controller-CNN pair (image part of layer 0 computed once,
  canvas part per timestep) against the plain N-packed pair and the oracle."""
  B, H, W = shape
  rng = np.random.RandomState(11)
  img = rng.rand(B, H, W, 4).astype(np.float32)
  canvas = rng.rand(B, H, W).astype(np.float32)
  wA, wB = (rng.randn(3, 3, 4, 8) * 0.3).astype(np.float32), (rng.randn(3, 3, 8, 8) * 0.2).astype(np.float32)
  scA, shA = rng.uniform(0.5, 1.5, 16).astype(np.float32), rng.randn(16).astype(np.float32) * 0.1
  scB, shB = rng.uniform(0.5, 1.5, 16).astype(np.float32), rng.randn(16).astype(np.float32) * 0.1
  wpA, wpB = dev(ops.pack_conv_weights(wA), cuda), dev(ops.pack_conv_weights(wB), cuda)
  d = lambda a: dev(a, cuda)
  plain = ops.conv_pair(d(img), wpA, d(scA), d(shA), 8, wpB, d(scB), d(shB), 8, poolB=2, plane=d(canvas), plane_chan=3)
  assert ops.first_cache_supported(4, 8, 8, 2, H, W)
  cache = ops.first_cache_alloc(B, H, W, cuda)
  ops.first_cache(d(img), wpA, 8, 3, cache)
  out = torch.empty_like(plain)
  ops.conv_pair_cached(cache, d(canvas), 3, wpA, d(scA), d(shA), wpB, d(scB), d(shB), 8, out)
  x = img.copy()
  x[..., 3] = canvas
  ref = ora.conv2d(x.astype(np.float64), wA.astype(np.float64)) * scA[:8] + shA[:8]
  ref = ora.conv2d(np.maximum(ref, 0), wB.astype(np.float64)) * scB[:8] + shB[:8]
  ref = ora.max_pool(np.maximum(ref, 0), 2)
  assert np.abs(plain.cpu().numpy() - ref).max() < 1e-4
  assert np.abs(out.cpu().numpy() - ref).max() < 1e-4
  # the first timestep's form: zero canvas, the plain kernel writes the cache as a by-product
  zero = torch.zeros_like(d(canvas))
  cache2 = ops.first_cache_alloc(B, H, W, cuda)
  out0 = torch.empty_like(plain)
  ops.conv_pair_fill_cache(d(img), zero, 3, wpA, d(scA), d(shA), wpB, d(scB), d(shB), 8, cache2, out0)
  plain0 = ops.conv_pair(d(img), wpA, d(scA), d(shA), 8, wpB, d(scB), d(shB), 8, poolB=2, plane=zero, plane_chan=3)
  assert np.abs(out0.cpu().numpy() - plain0.cpu().numpy()).max() < 1e-4
  out1 = torch.empty_like(plain)
  ops.conv_pair_cached(cache2, d(canvas), 3, wpA, d(scA), d(shA), wpB, d(scB), d(shB), 8, out1)
  assert np.abs(out1.cpu().numpy() - ref).max() < 1e-4   # the by-product cache serves the later timesteps
  # the same launch with the decode loop's y_out prefill riding on it: a constant written by the conv kernel's
  # workgroups (sizes that do not divide by the grid, guard words either side), results unchanged bit for bit
  for nfill in (4, 1024 * 1024 + 12, 3 * 256 * 4 * 7):
    buf = torch.full((nfill + 8,), 7.0, dtype=torch.float32, device=cuda)
    cache3, out3 = ops.first_cache_alloc(B, H, W, cuda), torch.empty_like(plain)
    ops.conv_pair_fill_cache(d(img), zero, 3, wpA, d(scA), d(shA), wpB, d(scB), d(shB), 8, cache3, out3,
                             fill=buf[4:4 + nfill], fill_value=0.25)
    got = buf.cpu().numpy()
    assert (got[:4] == 7.0).all() and (got[4 + nfill:] == 7.0).all() and (got[4:4 + nfill] == 0.25).all()
    assert (out3.cpu().numpy() == out0.cpu().numpy()).all() and (cache3.cpu().numpy() == cache2.cpu().numpy()).all()


@pytest.mark.parametrize('Dd,Dy,Cp', [(8, 2, 16), (8, 9, 24), (0, 0, 4), (8, 1, 16)])
def test_pack_input_zeroes_canvas_plane(cuda, Dd, Dy, Cp):
  """pack_input with the decode loop's separate canvas plane: the packed record [x | 0 | d_in | y_in | 0-pad] (full_model.py:640-661
  order, canvas slot zero) for the KITTI (13 in 16), Cityscapes (21 in 24) and CVPPP (3 + 1 in 4) inputs, plane = 0 (round 6: one
  thread per channel quad, 16-byte stores)."""
  rng = np.random.RandomState(5 + Cp)
  B, H, W = 2, 12, 20
  x = rng.rand(B, H, W, 3)
  d_in = rng.rand(B, H, W, Dd) if Dd else None
  y_in = rng.rand(B, H, W, Dy) if Dy else None
  packed = torch.full((B, H, W, Cp), -1.0, dtype=torch.float32, device=cuda)
  plane = torch.full((B, H, W), 3.0, dtype=torch.float32, device=cuda)
  ops.pack_input(dev(x, cuda), None if d_in is None else dev(d_in, cuda), None if y_in is None else dev(y_in, cuda), Cp, packed, canvas_plane=plane)
  got = packed.cpu().numpy()
  parts = [x, np.zeros((B, H, W, 1))] + ([d_in] if Dd else []) + ([y_in] if Dy else [])
  parts.append(np.zeros((B, H, W, Cp - 4 - Dd - Dy)))
  want = np.concatenate(parts, axis=-1).astype(np.float32)
  assert (got == want).all() and (plane.cpu().numpy() == 0).all()


def test_adam_step_matches_tf_adam(cuda):
  """ra_adam_step_f32: clip(g / world + wd * w, +-1) then tf.train.AdamOptimizer's update
  (full_model.py:1046-1056), three steps against a float64 restatement."""
  import ctypes as C
  rng = np.random.RandomState(0)
  n = 1003
  p0, wd = rng.randn(n).astype(np.float32), (rng.rand(n) < 0.5).astype(np.float32) * 5e-5
  p, m, v = dev(p0, cuda), dev(np.zeros(n), cuda), dev(np.zeros(n), cuda)
  pr, mr, vr = p0.astype(np.float64), np.zeros(n), np.zeros(n)
  b1, b2, eps, lr, world = 0.9, 0.999, 1e-7, 1e-3, 2
  for t in range(1, 4):
    g = (rng.randn(n) * 3).astype(np.float32)  # many elements beyond the clip
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    gd, wdd = dev(g, cuda), dev(wd, cuda)  # keep the device copies alive across the launch
    rn.check(rn.lib().ra_adam_step_f32(rn.ptr(p), rn.ptr(gd), rn.ptr(m), rn.ptr(v), rn.ptr(wdd),
                                       n, C.c_float(lr_t), C.c_float(b1), C.c_float(b2), C.c_float(eps),
                                       C.c_float(1.0), C.c_float(1.0 / world), rn.stream_ptr()), 'adam')
    gg = np.clip(g.astype(np.float64) / world + wd * pr, -1, 1)
    mr = b1 * mr + (1 - b1) * gg
    vr = b2 * vr + (1 - b2) * gg * gg
    pr = pr - lr_t * mr / (np.sqrt(vr) + eps)
    assert np.abs(p.cpu().numpy() - pr).max() < 1e-5
    del gd, wdd


@pytest.mark.parametrize('B,H,W,Ci,Co,pool,relu', [
    (2, 32, 48, 32, 32, 2, True),    # controller CNN L5 shape class
    (3, 16, 16, 32, 64, 2, True),    # L6: two cout slices
    (2, 48, 32, 16, 32, 1, True),    # L4: no pool, Cin 16
    (1, 32, 32, 16, 32, 2, False),
    (2, 64, 32, 16, 16, 2, True),    # L3 shape class: one block of 16 output channels per workgroup
    (1, 16, 48, 32, 16, 1, True),
    (8, 128, 128, 32, 32, 2, True),  # cfg2's L5 itself: persistent workgroups, several tiles each
])
def test_conv_winograd(cuda, B, H, W, Ci, Co, pool, relu):
  """ra_conv_wino_f32 (Winograd F(2x2,3x3) on the MFMA) against the float64 direct convolution and
  against ra_conv3x3_f32 on the same inputs."""
  rng = np.random.RandomState(B * 100 + H + Ci + Co + pool)
  x = rng.randn(B, H, W, Ci).astype(np.float32)
  w = (rng.randn(3, 3, Ci, Co) / np.sqrt(9 * Ci)).astype(np.float32)
  b = rng.randn(Co).astype(np.float32) * 0.1
  bn = tuple(a.astype(np.float32) for a in (rng.randn(Co) * 0.2, rng.uniform(0.5, 1.5, Co) * rng.choice([-1, 1], Co),
                                            rng.randn(Co) * 0.2, rng.uniform(0.5, 1.5, Co)))
  assert ops.conv_wino_supported(Ci, Co, pool, H, W)
  assert not ops.conv_wino_supported(Ci, Co, pool, H + 8, W) and not ops.conv_wino_supported(8, Co, pool, H, W)
  sc, sh = ops.fold_bn(b, Co, bn)
  xd, scd, shd = dev(x, cuda), dev(sc, cuda), dev(sh, cuda)
  ops.poison_lds()
  y = ops.conv_wino(xd, dev(ops.pack_wino_weights(w), cuda), scd, shd, Co, relu=relu, pool=pool)
  direct = ops.conv3x3(xd, dev(ops.pack_conv_weights(w), cuda), scd, shd, Co, relu=relu, pool=pool)
  torch.cuda.synchronize()
  y, direct = y.cpu().numpy(), direct.cpu().numpy()
  assert y.shape == direct.shape
  assert relerr(y, direct) < 2e-5
  if B * H * W <= 8192:
    ref = ora.batch_norm_eval(ora.conv2d(x.astype(np.float64), w.astype(np.float64)) + b, *[a.astype(np.float64) for a in bn])
    ref = ora.relu(ref) if relu else ref
    ref = ora.max_pool(ref, pool) if pool > 1 else ref
    assert relerr(y, ref) < 2e-5


@pytest.mark.parametrize('B,H,W', [(2, 32, 48), (1, 16, 16), (8, 256, 256)])
def test_conv_pair_winograd(cuda, B, H, W):
  _pair_winograd_case(cuda, B, H, W)


def _pair_winograd_case(cuda, B, H, W):
  """ra_conv_pair_wino_f32 (8 -> 16 -> 16, pool 2; layer B as Winograd from the LDS tile) against the
  direct fused pair on the same inputs and, at the small sizes, the float64 oracle."""
  rng = np.random.RandomState(B + H + W)
  x = rng.randn(B, H, W, 8).astype(np.float32)
  wA = (rng.randn(3, 3, 8, 16) / np.sqrt(72)).astype(np.float32)
  wB = (rng.randn(3, 3, 16, 16) / np.sqrt(144)).astype(np.float32)
  bA, bB = rng.randn(16).astype(np.float32) * 0.1, rng.randn(16).astype(np.float32) * 0.1
  mk = lambda: tuple(a.astype(np.float32) for a in (rng.randn(16) * 0.2, rng.uniform(0.5, 1.5, 16) * rng.choice([-1, 1], 16),
                                                     rng.randn(16) * 0.2, rng.uniform(0.5, 1.5, 16)))
  bnA, bnB = mk(), mk()
  scA, shA = [dev(a, cuda) for a in ops.fold_bn(bA, 16, bnA)]
  scB, shB = [dev(a, cuda) for a in ops.fold_bn(bB, 16, bnB)]
  assert ops.conv_pair_wino_supported(8, 16, 16, 2, H, W) and not ops.conv_pair_wino_supported(8, 16, 16, 1, H, W)
  xd, wpa = dev(x, cuda), dev(ops.pack_conv_weights(wA), cuda)
  ops.poison_lds()
  y = ops.conv_pair_wino(xd, wpa, scA, shA, dev(ops.pack_wino_weights(wB), cuda), scB, shB)
  direct = ops.conv_pair(xd, wpa, scA, shA, 16, dev(ops.pack_conv_weights(wB), cuda), scB, shB, 16, poolB=2)
  torch.cuda.synchronize()
  y, direct = y.cpu().numpy(), direct.cpu().numpy()
  assert y.shape == direct.shape == (B, H // 2, W // 2, 16)
  assert relerr(y, direct) < 2e-5
  if B * H * W <= 8192:
    h = ora.relu(ora.batch_norm_eval(ora.conv2d(x.astype(np.float64), wA.astype(np.float64)) + bA, *[a.astype(np.float64) for a in bnA]))
    ref = ora.relu(ora.batch_norm_eval(ora.conv2d(h, wB.astype(np.float64)) + bB, *[a.astype(np.float64) for a in bnB]))
    assert relerr(y, ora.max_pool(ref, 2)) < 2e-5
