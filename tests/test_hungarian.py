"""Hungarian matching: the C oracle and the product (host + device entry points) against the
reference's own known-answer tests (hungarian_tf_tests.py:9-90, fixture
tests/golden/hungarian_kats.json), its termination set (:92-275), and each other (bit-exact)."""
import ctypes
import json
import os

import numpy as np
import pytest

import ra_ops as ops

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, 'golden', 'hungarian_kats.json')))['cases']
_ora = ctypes.CDLL(os.path.join(os.path.dirname(HERE), 'oracle', 'libhungarian_oracle.so'))
_ora.ora_hungarian_f32.restype = ctypes.c_int


def oracle(W):
  W = np.ascontiguousarray(W, np.float32)
  two_d = W.ndim == 2
  W3 = W[None] if two_d else W
  B, N, M = W3.shape
  m, cx, cy = np.zeros_like(W3), np.zeros((B, N), np.float32), np.zeros((B, M), np.float32)
  p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
  rc = _ora.ora_hungarian_f32(p(W3), B, N, M, p(m), p(cx), p(cy))
  return rc, m, cx, cy


def _weights(case):
  W = np.array(case['W'], dtype=np.float64)
  if case['round_1e6']:
    W = np.round(W * 1e6) / 1e6  # hungarian_tf_tests.py:200-201
  return W.astype(np.float32)  # the op's input is float32 (hungarian.cc:27)


@pytest.mark.parametrize('case', KATS, ids=[c['name'] for c in KATS])
def test_reference_vectors_oracle_and_host(case):
  W = _weights(case)
  rc, m, cx, cy = oracle(W)
  M2, cx2, cy2 = ops.hungarian(W)
  assert rc == 0 and ops.hungarian.last_status == 0  # terminates without hitting a cap
  if W.ndim == 2:
    m, cx, cy = m[0], cx[0], cy[0]
  # product == oracle, bit for bit
  assert (M2 == m).all() and (cx2[..., 0] == cx).all() and (cy2[..., 0, :] == cy).all()
  # and both == the reference's asserted answers where it asserts them
  if 'matching' in case:
    assert (m == np.array(case['matching'], np.float32)).all()
  if 'cover_x' in case:
    assert (cx == np.array(case['cover_x'], np.float32)).all()
    assert (cy == np.array(case['cover_y'], np.float32)).all()
  # a matching: at most one 1 per row/column, entries exactly 0/1
  assert set(np.unique(m)) <= {0.0, 1.0}
  assert (m.sum(-1) <= 1).all() and (m.sum(-2) <= 1).all()


def _random_cases(seed, n):
  rng = np.random.RandomState(seed)
  for it in range(n):
    nx = rng.randint(1, 24)
    ny = nx if rng.rand() < 0.6 else rng.randint(1, 24)
    W = rng.rand(rng.randint(1, 4), nx, ny).astype(np.float32)
    kind = it % 5
    if kind == 1:   # the f_segm_match conditioning (modellib.py:395-406)
      W = (np.floor(W * 1e6 + 0.5) / 1e6 + 1e-5).astype(np.float32)
    elif kind == 2:  # masked-out GT columns
      W[:, :, ny // 2:] = 1e-5
    elif kind == 3:  # ties
      W = np.round(W * 4) / 4
    elif kind == 4:  # zeros
      W[W < 0.5] = 0
    yield W


def test_host_equals_oracle_random():
  for W in _random_cases(0, 250):
    a = oracle(W)
    M, cx, cy = ops.hungarian(W)
    assert a[0] == ops.hungarian.last_status
    assert (a[1] == M).all() and (a[2] == cx[..., 0]).all() and (a[3] == cy[:, 0, :]).all()


def test_optimality_small():
  """The matching maximises total weight (brute force over permutations, n <= 6)."""
  import itertools
  rng = np.random.RandomState(3)
  for _ in range(40):
    n = rng.randint(2, 7)
    W = (rng.randint(1, 50, (n, n))).astype(np.float32)
    M, _, _ = ops.hungarian(W)
    best = max(sum(W[i, p[i]] for i in range(n)) for p in itertools.permutations(range(n)))
    assert (M * W).sum() == best


def test_shape_and_rank_errors():
  with pytest.raises(Exception):
    ops.hungarian(np.zeros((2, 2, 2, 2), np.float32))
  M, cx, cy = ops.hungarian(np.ones((3, 4), np.float32))
  assert M.shape == (3, 4) and cx.shape == (3, 1) and cy.shape == (1, 4)
  M, cx, cy = ops.hungarian(np.ones((2, 3, 4), np.float32))
  assert M.shape == (2, 3, 4) and cx.shape == (2, 3, 1) and cy.shape == (2, 1, 4)


def test_f_segm_match_conditioning():
  """modellib.f_segm_match (modellib.py:382-415) on CPU tensors: masks + quantisation."""
  import torch
  import modellib
  rng = np.random.RandomState(1)
  iou = torch.from_numpy(rng.rand(2, 5, 5).astype(np.float32))
  s_gt = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]], dtype=torch.float32)
  m = modellib.f_segm_match(iou, s_gt).numpy()
  assert (m[0, 3:, :] == 0).all() and (m[0, :, 3:] == 0).all()
  assert m[0].sum() == 3 and m[1].sum() == 5


@pytest.mark.gpu
def test_device_equals_host(cuda):
  import torch
  for W in list(_random_cases(5, 60)) + [_weights(c) for c in KATS]:
    Mh, cxh, cyh = ops.hungarian(W)
    st_h = ops.hungarian.last_status
    Md, cxd, cyd = ops.hungarian(torch.from_numpy(W).to(cuda))
    torch.cuda.synchronize()
    assert (Md.cpu().numpy() == Mh).all()
    assert (cxd.cpu().numpy() == cxh).all() and (cyd.cpu().numpy() == cyh).all()
    assert int(ops.hungarian.last_status.max()) == st_h
  # cfg5-sized problem: T = 32
  W = np.random.RandomState(9).rand(4, 32, 32).astype(np.float32)
  Mh, _, _ = ops.hungarian(W)
  Md, _, _ = ops.hungarian(torch.from_numpy(W).to(cuda))
  assert (Md.cpu().numpy() == Mh).all()


@pytest.mark.gpu
@pytest.mark.parametrize('ring,bfs', [('0', '1'), ('64', '1'), ('64', '0'), ('8192', '0')])
def test_device_queue_ring_spill(cuda, monkeypatch, ring, bfs):
  """The BFS queue's LDS ring is only a cache of the global queue: with no ring, and with one so
  small that the live window overruns it mid-search, the device result is still the host's — for
  the chunk-parallel search (bfs = 1: a round that would overrun the ring restarts one pop at a time)
  and for the one-pop-at-a-time search (bfs = 0)."""
  import torch
  monkeypatch.setenv('RA_HUNG_RING', ring)
  monkeypatch.setenv('RA_HUNG_BFS', bfs)
  rng = np.random.RandomState(21)
  iou = rng.rand(8, 21, 21).astype(np.float32)
  iou[:, :, 13:] = 0  # f_segm_match's shape: dead ground-truth columns all at the eps fill
  for W in (iou + 1e-5, rng.rand(4, 32, 32).astype(np.float32)):
    Mh, cxh, cyh = ops.hungarian(W)
    Md, cxd, cyd = ops.hungarian(torch.from_numpy(W).to(cuda))
    assert (Md.cpu().numpy() == Mh).all()
    assert (cxd.cpu().numpy() == cxh).all() and (cyd.cpu().numpy() == cyh).all()


@pytest.mark.gpu
@pytest.mark.parametrize('case', KATS, ids=[c['name'] for c in KATS])
def test_reference_vectors_device(cuda, case):
  """The DEVICE solver against the reference's own asserted answers (hungarian_tf_tests.py:9-90)
  and, bit for bit, against the C oracle — not only transitively through the host product."""
  import torch
  W = _weights(case)
  rc, m, cx, cy = oracle(W)
  Md, cxd, cyd = ops.hungarian(torch.from_numpy(W).to(cuda))
  assert int(ops.hungarian.last_status.max()) == 0 and rc == 0
  Md, cxd, cyd = Md.cpu().numpy(), cxd.cpu().numpy()[..., 0], cyd.cpu().numpy()[..., 0, :]
  if W.ndim == 2:
    m, cx, cy = m[0], cx[0], cy[0]
  assert (Md == m).all() and (cxd == cx).all() and (cyd == cy).all()
  if 'matching' in case:
    assert (Md == np.array(case['matching'], np.float32)).all()
  if 'cover_x' in case:
    assert (cxd == np.array(case['cover_x'], np.float32)).all()
    assert (cyd == np.array(case['cover_y'], np.float32)).all()


@pytest.mark.gpu
def test_status_is_checked_on_device(cuda):
  """A per-example inner-cap status must surface as an error, as the reference's LOG(FATAL) does
  (hungarian.cc:124-127): a near-dense equality graph at T = 32 exceeds the BFS pop cap."""
  import torch
  from ra_native import RecAttendError
  W = np.full((1, 32, 32), 1e-5, np.float32)  # one image without any GT object, f_segm_match's eps fill
  rc, _, _, _ = oracle(W)
  if rc < 0:  # the reference aborts here; so must the product, host and device alike
    with pytest.raises(RecAttendError):
      ops.hungarian(W)
    with pytest.raises(RecAttendError):
      ops.hungarian(torch.from_numpy(W).to(cuda))
  else:
    M, _, _ = ops.hungarian(torch.from_numpy(W).to(cuda))
    assert int(ops.hungarian.last_status.max()) == rc


# ---- the matrices a real cfg4 training step produces (tests/golden/hungarian_cfg4_step.npz, written by
# tests/golden/make_hungarian_step_fixture.py from a capture of tools/hungarian_step_probe.py): 16 problems of 16 x 16 with
# 8-15 live ground-truth columns and near-uniform IoUs — 1.1-1.8 ms each on the device solver, 50 x the planted-permutation
# matrices bench.py used to quote alone
STEP = np.load(os.path.join(HERE, 'golden', 'hungarian_cfg4_step.npz'))


def test_cfg4_step_matrices_oracle_and_host():
  import torch
  import modellib
  import ra_oracle as ora
  iou, s = STEP['iou'], STEP['s_gt']
  w, mx, my = ora.f_segm_match_precondition(iou, s)
  assert (w == STEP['weights']).all()
  rc, m, cx, cy = oracle(w)
  assert rc == 0 and (m * mx * my == STEP['match']).all()  # the committed answer is the oracle's
  M, cx2, cy2 = ops.hungarian(w)
  assert ops.hungarian.last_status == 0
  assert (M == m).all() and (cx2[..., 0] == cx).all() and (cy2[:, 0, :] == cy).all()  # host product == oracle, bit for bit
  got = modellib.f_segm_match(torch.from_numpy(iou), torch.from_numpy(s)).numpy()     # through the reference-named caller
  assert (got == STEP['match']).all()
  assert (STEP['match'].sum((1, 2)) == s.sum(1)).all()  # every live instance is matched


@pytest.mark.gpu
def test_cfg4_step_matrices_device(cuda):
  import torch
  w = torch.from_numpy(STEP['weights']).to(cuda)
  Md, cxd, cyd = ops.hungarian(w)
  assert int(ops.hungarian.last_status.max()) == 0
  rc, m, cx, cy = oracle(STEP['weights'])
  assert (Md.cpu().numpy() == m).all() and (cxd.cpu().numpy()[..., 0] == cx).all() and (cyd.cpu().numpy()[:, 0, :] == cy).all()
  match, status = ops.segm_match(torch.from_numpy(STEP['iou']).to(cuda), torch.from_numpy(STEP['s_gt']).to(cuda))
  assert int(status.abs().max()) == 0 and (match.cpu().numpy() == STEP['match']).all()


@pytest.mark.gpu
def test_segm_match_on_host_cores_as_a_stream_host_function(cuda):
  """ra_segm_match_host_f32 (round 6): f_segm_match with the Hungarian problems solved by ra_hungarian_f32 on host threads, side by
  side, as a host function of the stream — precondition kernel, D2H into the caller's pinned block, host solve, H2D, re-mask.  The
  cfg4 step's matrices against the committed answer, random ragged problems against the device solver (bit for bit, statuses
  included), the same pinned block reused, and the call captured in a HIP graph and replayed on new inputs (the training step's use)."""
  import time
  import torch
  iou = torch.from_numpy(STEP['iou']).to(cuda)
  s_gt = torch.from_numpy(STEP['s_gt']).to(cuda)
  match, status, block = ops.segm_match_host(iou, s_gt, None, threads=16)
  torch.cuda.synchronize()
  assert int(status.abs().max()) == 0 and (match.cpu().numpy() == STEP['match']).all()
  rng = np.random.RandomState(11)
  for B, N in ((16, 16), (5, 21), (32, 16), (3, 32)):
    iou_r = rng.rand(B, N, N).astype(np.float32) * (rng.rand(B, N, N) < 0.7)
    s_r = (np.arange(N)[None, :] < rng.randint(1, N + 1, (B, 1))).astype(np.float32)
    a, b = torch.from_numpy(iou_r).to(cuda), torch.from_numpy(s_r).to(cuda)
    md, sd = ops.segm_match(a, b)
    mh, sh, blk = ops.segm_match_host(a, b, None, threads=3)
    mh2, sh2, blk2 = ops.segm_match_host(a, b, blk, threads=64)  # the block reused, more threads than problems
    torch.cuda.synchronize()
    assert blk2 is blk and torch.equal(md, mh) and torch.equal(sd, sh) and torch.equal(md, mh2) and torch.equal(sd, sh2)
  # under graph capture: static inputs, new values before every replay
  st_iou, st_s = iou.clone(), s_gt.clone()
  ops.segm_match_host(st_iou, st_s, block, threads=16)
  torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    gm, gs, gb = ops.segm_match_host(st_iou, st_s, block, threads=16)
  for k in range(3):
    perm = torch.randperm(16, generator=torch.Generator().manual_seed(k)).to(cuda)
    st_iou.copy_(iou[perm])
    st_s.copy_(s_gt[perm])
    g.replay()
    torch.cuda.synchronize()
    assert (gm.cpu().numpy() == STEP['match'][perm.cpu().numpy()]).all() and int(gs.abs().max()) == 0
  t0 = time.perf_counter()
  for _ in range(5):
    g.replay()
  torch.cuda.synchronize()
  t_host = (time.perf_counter() - t0) / 5
  ops.segm_match(iou, s_gt)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(5):
    ops.segm_match(iou, s_gt)
  torch.cuda.synchronize()
  print('cfg4 step matchings: host node %.2f ms, device solver %.2f ms' % (1e3 * t_host, 1e3 * (time.perf_counter() - t0) / 5))
