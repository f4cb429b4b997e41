"""Dynamic tile tickets (ra_tile_tickets_bind, include/recattend.h; csrc/ra_common.h TicketWalk): the persistent controller-CNN
launches with DRAWN tiles give bit for bit what the static tile walk gives — alone, with the slots of one scratch used launch
after launch, and while another stream keeps taking the CUs away (late workgroups must neither skip nor repeat a tile)."""
import numpy as np
import pytest
import torch

import ra_native as rn
import ra_ops as ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def cuda():
  if not torch.cuda.is_available():
    pytest.skip('needs an MI355X')
  return torch.device('cuda')


def dev(a, cuda):
  return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _cases(cuda):
  """name -> (closure that runs the op and returns its output, tiles of the launch)"""
  rng = np.random.RandomState(5)
  d = lambda a: dev(a.astype(np.float32), cuda)
  out = {}
  # first controller-CNN pair, cached form (conv_pair8_mfma<4, CACHED, SPLIT>): 16 x 32 tiles
  B, H, W = 24, 256, 256  # 3072 tiles on 768 workgroups (tiles are drawn from 3 per workgroup upwards)
  img, canvas = d(rng.rand(B, H, W, 4)), d(rng.rand(B, H, W))
  wA, wB = (rng.randn(3, 3, 4, 8) * 0.3).astype(np.float32), (rng.randn(3, 3, 8, 8) * 0.2).astype(np.float32)
  scA, shA, scB, shB = d(rng.uniform(0.5, 1.5, 16)), d(rng.randn(16) * 0.1), d(rng.uniform(0.5, 1.5, 16)), d(rng.randn(16) * 0.1)
  wpA, wpB = dev(ops.pack_conv_weights(wA), cuda), dev(ops.pack_conv_weights(wB), cuda)
  cache = ops.first_cache_alloc(B, H, W, cuda)
  ops.first_cache(img, wpA, 8, 3, cache)
  o8 = torch.empty((B, H // 2, W // 2, 8), dtype=torch.float32, device=cuda)

  def pair8():
    ops.conv_pair_cached(cache, canvas, 3, wpA, scA, shA, wpB, scB, shB, 8, o8)
    return o8
  out['pair8'] = (pair8, (H // 16) * (W // 32) * B)
  # second pair with its Winograd layer B (conv_pair_wino_mfma<8, SPLIT>): 8 x 16 tiles
  x8 = d(rng.randn(16, 256, 256, 8))
  wA2, wB2 = (rng.randn(3, 3, 8, 16) / np.sqrt(72)).astype(np.float32), (rng.randn(3, 3, 16, 16) / 12).astype(np.float32)
  wpa2, wpb2 = dev(ops.pack_conv_weights(wA2), cuda), dev(ops.pack_wino_weights(wB2), cuda)
  out['pair_wino'] = (lambda: ops.conv_pair_wino(x8, wpa2, scA, shA, wpb2, scB, shB), 32 * 16 * 16)
  # K1w (conv_wino_mfma): 16 -> 32 at 128 x 128, 16 x 16 tiles
  x16 = d(rng.randn(48, 128, 128, 16))
  w3 = (rng.randn(3, 3, 16, 32) / 12).astype(np.float32)
  sc32, sh32 = d(rng.uniform(0.5, 1.5, 32)), d(rng.randn(32) * 0.1)
  wp3 = dev(ops.pack_wino_weights(w3), cuda)
  out['wino'] = (lambda: ops.conv_wino(x16, wp3, sc32, sh32, 32, relu=True, pool=1), 64 * 48)
  # K1s (conv_split_kernel): 32 -> 32 (one cout slice) and 32 -> 64 (two slices, each with its own pools)
  x32 = d(rng.randn(16, 128, 128, 32))
  for co in (32, 64):
    w4 = (rng.randn(3, 3, 32, co) / 17).astype(np.float32)
    wp4 = torch.from_numpy(ops.pack_split_weights(w4)).to(cuda)
    sc, sh = d(rng.uniform(0.5, 1.5, co)), d(rng.randn(co) * 0.1)
    out['split%d' % co] = (lambda wp4=wp4, sc=sc, sh=sh, co=co: ops.conv_split(x32, wp4, sc, sh, co, relu=True, pool=2), 64 * 16)
  return out


def test_drawn_tiles_equal_the_static_walk(cuda):
  slot_words = rn.lib().ra_tile_tickets_slot_bytes() // 4
  for name, (run, ntiles) in _cases(cuda).items():
    ref = run().clone()
    torch.cuda.synchronize()
    tk = ops.tickets_alloc(8, cuda)
    if not ops.tickets_bind(tk):
      pytest.skip('tile tickets are not available on this device (XCC census)')
    try:
      got = [run().clone() for _ in range(3)]  # three launches, three (or six) fresh slots of the same scratch
    finally:
      ops.tickets_unbind()
    torch.cuda.synchronize()
    for g in got:
      assert torch.equal(g, ref), name
    counters = tk.view(torch.int32).cpu().numpy().reshape(8, 8, slot_words // 8)[:, :, 0]  # [slot][XCD pool]
    nsl = 2 if name == 'split64' else 1
    for k in range(3 * nsl):
      # every workgroup draws its tiles + 2: the pools of a slot hand out all tiles, and the draws beyond them are two per workgroup
      assert counters[k].sum() > ntiles and (counters[k].sum() - ntiles) % 2 == 0, (name, k, counters[k])
      assert (counters[k] >= (ntiles + 7) // 8).all(), (name, k, counters[k])  # every pool was drained
    assert (counters[3 * nsl:] == 0).all(), name
    # unbound again: the static walk, the scratch untouched
    before = tk.clone()
    assert torch.equal(run(), ref) and torch.equal(tk, before)


def test_drawn_tiles_with_company_on_the_gpu(cuda):
  """Another stream launches kernels that claim the whole LDS of every CU (ra_debug_poison_lds: 1024 workgroups of 160 KB)
  while the ticketed launches run: their workgroups start late, in any order — every tile is still computed exactly once."""
  side = torch.cuda.Stream()
  for name, (run, ntiles) in _cases(cuda).items():
    ref = run().clone()
    torch.cuda.synchronize()
    tk = ops.tickets_alloc(16, cuda)
    for rep in range(2):
      if not ops.tickets_bind(tk):
        pytest.skip('tile tickets are not available on this device (XCC census)')
      try:
        with torch.cuda.stream(side):
          for _ in range(12):
            ops.poison_lds()
        got = [run().clone() for _ in range(4)]
      finally:
        ops.tickets_unbind()
      torch.cuda.synchronize()
      for g in got:
        assert torch.equal(g, ref), (name, rep)


def test_forward_with_drawn_tiles_is_bit_identical(cuda):
  """cfg2-shaped forward (512 x 512, B = 8, the captured graph) with DecodeEngine.tile_tickets on and off: same y_out, s_out,
  canvas bit for bit, replay after replay (the ticket scratch is re-zeroed inside the graph), and the scratch shows that the
  first pair's launches did draw (5.3 tiles per workgroup there)."""
  import full_model
  import ra_oracle as ora
  opt = ora.make_opt('cvppp', 512, 512, 4)
  P = ora.random_params(opt, 101)
  x = torch.rand((8, 512, 512, 3), device=cuda)
  outs = {}
  for on in (False, True):
    m = full_model.get_model(opt).load_weights(P)
    m.engine.tile_tickets = on
    runs = []
    for _ in range(3):
      y, s = m.run(['y_out', 's_out'], {'x': x, 'phase_train': False})
      runs.append((torch.as_tensor(y).clone(), torch.as_tensor(s).clone()))
    for y, s in runs[1:]:
      assert torch.equal(y, runs[0][0]) and torch.equal(s, runs[0][1])
    outs[on] = runs[0]
    if on:
      tk = m.engine.subs[0]['tickets'].view(torch.int32)
      if not ops.tickets_bind(m.engine.subs[0]['tickets'].clone()):
        pytest.skip('tile tickets are not available on this device (XCC census)')
      ops.tickets_unbind()
      assert int((tk != 0).sum()) > 0
  assert torch.equal(outs[True][0], outs[False][0]) and torch.equal(outs[True][1], outs[False][1])
