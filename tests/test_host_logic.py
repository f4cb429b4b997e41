"""Host-side logic that needs no GPU: the model_opt -> shape bookkeeping, the weight-key schema,
the kernel weight packing (checked by index arithmetic), error behaviour."""
import ctypes as C

import numpy as np
import pytest
import torch

import full_model
import box_model
import nnlib as nn
import modellib
import ra_ops as ops
import ra_native as rn
import ra_oracle as ora


def test_dims_match_oracle_bookkeeping():
  for arch, H, W, T, over in (('cvppp', 128, 128, 5, {}), ('kitti', 128, 448, 20, {}),
                              ('cityscapes', 256, 512, 20, {}),
                              ('cvppp', 512, 512, 16, {'fixed_var': True})):
    opt = ora.make_opt(arch, H, W, T, **over)
    d, o = full_model.derive_dims(opt), ora.derive(opt)
    for k in ('G', 'gh', 'gw', 'core_dim', 'ccnn_channels', 'acnn_channels', 'adcnn_channels',
              'ctrl_in', 'attn_in', 'fixed_var', 'dynamic_var', 'fixed_gamma', 'disable_overwrite'):
      assert d[k] == o[k], (arch, k)
    assert (d['skip_ch'] or [])[:d['adcnn_nlayers']] == (o['skip_ch'] or [])[:d['adcnn_nlayers']]
  assert full_model.derive_dims(ora.make_opt('kitti', 128, 448, 20))['C0p'] == 16
  assert full_model.derive_dims(ora.make_opt('cityscapes', 256, 512, 20))['C0p'] == 24


def test_weight_key_schema_is_the_reference_one():
  opt = ora.make_opt('cvppp', 64, 64, 3)
  m = full_model.get_model(opt)
  S = ora.param_shapes(opt)
  assert set(m.weight_keys()) == set(S)
  assert all(tuple(m[k].shape) == tuple(S[k]) for k in S)
  # spot-check the literal key forms (nnlib.py:124,208,334,472,612-623)
  for k in ('ctrl_cnn_w_0', 'ctrl_cnn_b_7', 'ctrl_cnn_0_2_beta', 'ctrl_cnn_7_0_ema_var',
            'ctrl_lstm_w_xi', 'ctrl_lstm_b_o', 'glimpse_mlp_w_1', 'ctrl_mlp_b_0', 'attn_cnn_w_5',
            'attn_cnn_5_2_gamma', 'attn_dcnn_w_6', 'attn_dcnn_6_1_ema_mean', 'score_mlp_w_0'):
    assert k in m
  assert tuple(m['attn_dcnn_w_0'].shape) == (3, 3, 32, 32)   # [f, f, out, in] (nnlib.py:321)
  assert tuple(m['score_mlp_w_0'].shape) == (256 + 6 * 6 * 32, 1)
  # reference initial values: BN gamma 1 / beta 0 / EMA 0, LSTM forget bias 1 (nnlib.py:89-91,567)
  assert float(m['ctrl_cnn_3_1_gamma'].mean()) == 1 and float(m['ctrl_cnn_3_1_ema_var'].sum()) == 0
  assert float(m['ctrl_lstm_b_f'].mean()) == 1 and float(m['ctrl_lstm_b_i'].abs().sum()) == 0
  assert float(m['ctrl_cnn_w_0'].abs().max()) <= 0.02 + 1e-6  # truncated normal, 2 sigma
  b = box_model.get_model(opt)
  assert set(b.weight_keys()) == set(ora.param_shapes(opt, box_model=True))
  # round trip through the weight archive form
  m2 = full_model.get_model(opt).load_weights(m.state_dict_numpy())
  assert all(torch.equal(m[k].cpu(), m2[k].cpu()) for k in S)


def test_conv_weight_packing_layout():
  rng = np.random.RandomState(0)
  for Ci, Co in ((4, 8), (8, 16), (16, 32), (32, 64), (24, 16)):
    w = rng.randn(3, 3, Ci, Co).astype(np.float32)
    p = ops.pack_conv_weights(w)
    cp = ops.cout_padded(Co)
    CK = 16 if Ci % 16 == 0 else 8 if Ci % 8 == 0 else 4
    p = p.reshape(Ci // CK, 9, CK // 4, 4, cp)
    for c in range(Ci):
      got = p[c // CK, :, (c % CK) // 4, c % 4, :Co]
      assert (got == w[:, :, c, :].reshape(9, Co)).all()
    assert (p[..., Co:] == 0).all()
  # transposed-conv filter [3,3,Cout,Cin]: taps flipped, in/out swapped, padded channels zero
  wt = rng.randn(3, 3, 8, 20).astype(np.float32)
  cmap = list(range(16)) + [16, 17, 18, 19] + [-1] * 4
  p = ops.pack_conv_weights(wt, cin_kernel=24, chan_map=cmap, transposed=True).reshape(3, 9, 2, 4, 16)
  for c in range(24):
    got = p[c // 8, :, (c % 8) // 4, c % 4, :8]
    exp = wt[::-1, ::-1, :, cmap[c]].reshape(9, 8) if cmap[c] >= 0 else np.zeros((9, 8))
    assert (got == exp).all()


def test_bn_fold_and_controller_packing():
  rng = np.random.RandomState(1)
  b, beta, gamma, mean, var = [rng.randn(8).astype(np.float32) for _ in range(5)]
  var = np.abs(var) + 0.1
  sc, sh = ops.fold_bn(b, 8, (beta, gamma, mean, var))
  x = rng.randn(5, 8).astype(np.float32)
  ref = ora.batch_norm_eval(x.astype(np.float64) + b, beta, gamma, mean, var)
  assert np.abs(x * sc[:8] + sh[:8] - ref).max() < 1e-5 and (sc[8:] == 1).all() and (sh[8:] == 0).all()
  sc, sh = ops.fold_bn(b, 8, None)
  assert (sc == 1).all() and (sh[:8] == b).all()
  desc = ops.make_ctrl_desc(49, 8, 16, 5, 2, 1, 16, 224, 224, 48, 48, 0, 0, 0, 1)
  lstm = {k + g: rng.randn(*s).astype(np.float32) for g in 'ifuo'
          for k, s in (('w_x', (8, 16)), ('w_h', (16, 16)))}
  lstm.update({'b_' + g: rng.randn(16).astype(np.float32) for g in 'ifuo'})
  gm = [(rng.randn(16, 16).astype(np.float32), rng.randn(16).astype(np.float32)),
        (rng.randn(16, 49).astype(np.float32), rng.randn(49).astype(np.float32))]
  cm = [(rng.randn(16, 9).astype(np.float32), rng.randn(9).astype(np.float32))]
  p = ops.pack_ctrl_weights(desc, lstm, gm, cm)
  W = p[:24 * 64].reshape(24, 64)
  for gi, g in enumerate('ifou'):  # packed gate order i, f, o, u
    assert (W[:8, gi * 16:(gi + 1) * 16] == lstm['w_x' + g]).all()
    assert (W[8:, gi * 16:(gi + 1) * 16] == lstm['w_h' + g]).all()
  off = 24 * 64 + 64 + 16 * 16 + 16
  g2 = p[off:off + 16 * 52].reshape(16, 52)  # G = 49 padded to 52 columns
  assert (g2[:, :49] == gm[1][0]).all() and (g2[:, 49:] == 0).all()


def test_unbuilt_parts_fail_loudly():
  opt = ora.make_opt('cvppp', 64, 64, 2)
  m = full_model.get_model(opt)
  with pytest.raises(rn.RecAttendError):  # the training graph needs the ground truth in the feed
    m.run(['loss', 'train_step'], {'x': np.zeros((1, 64, 64, 3), np.float32), 'phase_train': True})
  if not torch.cuda.is_available():
    zeros = lambda *s: np.zeros(s, np.float32)
    with pytest.raises(rn.RecAttendError):  # and an MI355X: the optimizer / conv kernels have no CPU form
      m.run(['loss', 'train_step'], {'x': zeros(1, 64, 64, 3), 'y_gt': zeros(1, 2, 64, 64), 's_gt': zeros(1, 2),
                                     'phase_train': True})
  with pytest.raises(KeyError):
    m.run('nonsense', {'x': None})
  with pytest.raises(NotImplementedError):
    modellib.f_sem_loss(None, None)  # fg_model's semantic loss: out of scope
  with pytest.raises(NotImplementedError):  # dead code in the reference (full_model.py:971,1016)
    modellib.f_match_loss(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 2), 2, modellib.f_bce)
  if not torch.cuda.is_available():
    with pytest.raises(rn.RecAttendError):   # no silent CPU fallback
      m.run('y_out', {'x': np.zeros((1, 64, 64, 3), np.float32), 'phase_train': False})
    with pytest.raises(rn.RecAttendError):
      nn.max_pool(torch.zeros(1, 4, 4, 4), 2)
  with pytest.raises(Exception):
    full_model.derive_dims(ora.make_opt('cvppp', 66, 64, 2))


def test_eval_and_loss_modules_have_no_cpu_fallback():
  """The reference-named evaluation / loss modules run on the HIP kernels only: CPU tensors are
  refused loudly, never computed by torch behind the caller's back."""
  import analysis
  import image_ops
  from utils import postprocess as pp
  y = torch.rand(1, 3, 8, 8)
  s = torch.rand(1, 3)
  if not torch.cuda.is_available():
    for call in (lambda: pp.postprocess(y, s, 0.5), lambda: pp.apply_one_label(y),
                 lambda: pp.remove_tiny(y, s, 10), lambda: analysis.f_iou_pairwise(y, y),
                 lambda: analysis.f_symmetric_best_dice({'y_out': y, 'y_gt': y, 's_out': s, 's_gt': s}),
                 lambda: modellib.f_iou(y, y, 3, pairwise=True), lambda: modellib.get_gt_box(y),
                 lambda: modellib.f_segm_match(torch.rand(1, 3, 3), s) if False else ops_segm(y, s),
                 lambda: image_ops.random_transformation(torch.rand(1, 8, 8, 3), 2, True)):
      with pytest.raises(rn.RecAttendError):
        call()
  # evaluation mode of the augmentation is the identity and needs no kernel
  x = torch.rand(1, 8, 8, 3)
  r = image_ops.random_transformation(x, 2, False, y=y)
  assert r['x'] is x and r['y'] is y
  if not torch.cuda.is_available():
    with pytest.raises(rn.RecAttendError):  # morph / upsample are device kernels since round 5: no CPU fallback either
      pp.morph(y)
  with pytest.raises(Exception):
    analysis.create_analyzer('nope')


def ops_segm(y, s):
  import ra_ops
  return ra_ops.segm_match(torch.rand(1, 3, 3), s)


def test_small_loss_functions_match_numpy():
  """f_iou_box / f_match_loss / f_huber / f_squared_err (modellib.py:206-238,440-530): [B,T,...]-sized
  bookkeeping, checked against a NumPy restatement."""
  rng = np.random.RandomState(2)
  tl_a, tl_b = rng.rand(2, 3, 2) * 10, rng.rand(2, 3, 2) * 10
  br_a, br_b = tl_a + 1 + rng.rand(2, 3, 2) * 8, tl_b + 1 + rng.rand(2, 3, 2) * 8
  t = lambda a: torch.tensor(a, dtype=torch.float64)
  got = modellib.f_iou_box(t(tl_a), t(br_a), t(tl_b), t(br_b)).numpy()
  iy = np.maximum(0, np.minimum(br_a[..., 0], br_b[..., 0]) - np.maximum(tl_a[..., 0], tl_b[..., 0]))
  ix = np.maximum(0, np.minimum(br_a[..., 1], br_b[..., 1]) - np.maximum(tl_a[..., 1], tl_b[..., 1]))
  area = lambda tl, br: (br[..., 0] - tl[..., 0]) * (br[..., 1] - tl[..., 1])
  assert np.abs(got - iy * ix / (area(tl_a, br_a) + area(tl_b, br_b) - iy * ix)).max() < 1e-12
  p, g = rng.randn(2, 3, 4) * 2, rng.randn(2, 3, 4)
  match = np.zeros((2, 3, 3))
  match[0, 0, 1] = match[0, 1, 0] = match[1, 2, 2] = 1
  for fn, ref in ((modellib.f_squared_err, lambda e: 0.5 * e * e),
                  (modellib.f_huber, lambda e: np.where(e <= 1, 0.5 * e * e, np.abs(e) - 0.5))):
    got = float(modellib.f_match_loss(t(p), t(g), t(match), 3, fn))
    exp = sum(sum(ref(p[b, i] - g[b, j]).sum() for i in range(3) for j in range(3) if match[b, i, j]) /
              max(match[b].sum(), 1) for b in range(2)) / 2 / 4
    assert abs(got - exp) < 1e-12


def test_decode_pipeline_host_contract():
  """DecodePipeline without a GPU: argument checks come first, then the loud no-CPU-fallback error."""
  import full_model
  from ra_native import RecAttendError
  import ra_oracle as ora
  m = full_model.get_model(ora.make_opt('cvppp', 32, 32, 2))
  with pytest.raises(ValueError):
    m.pipeline(0)
  pipe = m.pipeline(2)
  assert len(pipe) == 0 and not pipe.full()
  with pytest.raises(RecAttendError):
    pipe.collect()
  if not torch.cuda.is_available():
    with pytest.raises(RecAttendError):
      pipe.submit(['y_out'], {'x': np.zeros((1, 32, 32, 3), np.float32)})


def test_deferred_status_check_raises_on_flush():
  """TrainStep checks a step's solver statuses one step late, from a pinned host record (the optimizer kernel has already
  refused a failed step's update on the device); flush_status() checks the last record now: a negative matching code raises
  (hungarian.cc's LOG(FATAL) cases), another rank's failure flag raises, a timed-out controller workgroup RECOVERS (warns,
  switches the trainer to the one-workgroup controller, gives the skipped step's number back), clean records pass and are
  consumed."""
  import warnings
  import torch
  import ra_native as rn
  import ra_train as rt

  class Ev:
    def synchronize(self):
      pass

  class Bucket:
    global_step = 7

  class Stub:
    _seqc = None
    _status_pending = None
    seq_ctrl_split = True
    dropped = 0
    _check_status = rt.TrainStep._check_status
    flush_status = rt.TrainStep.flush_status

    def __init__(self):
      self.bucket, self.model = Bucket(), {}

    def _drop_captured_steps(self):
      self.dropped += 1

  s = Stub()
  s._status_pending = (Ev(), torch.zeros(17, dtype=torch.int32), 16, 1)
  s.flush_status()
  assert s._status_pending is None
  s.flush_status()  # nothing pending: a no-op
  bad = torch.zeros(17, dtype=torch.int32)
  bad[3] = -2
  s._status_pending = (Ev(), bad, 16, 1)
  with pytest.raises(rn.RecAttendError):
    s.flush_status()
  assert s._status_pending is None
  ctl = torch.zeros(18, dtype=torch.int32)
  ctl[16] = 1
  s._status_pending = (Ev(), ctl, 16, 1)
  with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    s.flush_status()
  assert any('one-workgroup controller' in str(x.message) for x in w)
  assert s.seq_ctrl_split is False and s.dropped == 1 and s.bucket.global_step == 6 and s.skipped_steps == 1
  other = torch.zeros(18, dtype=torch.int32)
  other[17] = 0x3f800000  # the all-reduced float flag 1.0 of another rank, as the optimizer kernel sees it
  s._status_pending = (Ev(), other, 16, 1)
  with pytest.raises(rn.RecAttendError):
    s.flush_status()
  # the record of a data-parallel step: [16 matching | 1 controller | 0 forced | 2 all-reduced flags] — the decision is the same
  # on every rank: ANOTHER rank's controller time-out makes this rank recover too instead of raising ...
  s2 = Stub()
  rec = torch.zeros(19, dtype=torch.int32)
  rec[18] = 0x3f800000  # flag 1: "a controller timed out on some rank" (this rank's own word, [16], is clean)
  s2._status_pending = (Ev(), rec, 16, 1, 0, 2)
  with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    assert s2._check_status(s2._status_pending) is True
  assert any('on another rank' in str(x.message) for x in w)
  assert s2.seq_ctrl_split is False and s2.dropped == 1 and s2.bucket.global_step == 6 and s2.skipped_steps == 1
  # ... another rank's failed MATCHING raises here as it does there ...
  rec = torch.zeros(19, dtype=torch.int32)
  rec[17] = 0x40000000  # two ranks' matchings failed
  with pytest.raises(rn.RecAttendError):
    s2._check_status((Ev(), rec, 16, 1, 0, 2))
  # ... and a step skipped on purpose (queued behind a timed-out one) only gives its number back, whatever else it says
  rec = torch.zeros(20, dtype=torch.int32)
  rec[16], rec[17], rec[19] = 1, 1, 0x3f800000
  assert s2._check_status((Ev(), rec, 16, 1, 1, 2)) is False
  assert s2.bucket.global_step == 5 and s2.skipped_steps == 2 and s2.dropped == 1


def test_weights_stamp_sees_every_kind_of_change():
  """DecodeEngine._weights_stamp (compared after every launch): full_model.Model counts the mutations of the dict itself, in-place
  edits show in the tensors' version counters; a plain dict as the model falls back to the (key, pointer, version) tuple."""
  import torch
  import full_model
  import ra_engine
  sys_opt = ora.make_opt('cvppp', 64, 64, 2)
  m = full_model.get_model(sys_opt)
  e = m.engine
  s0 = e._weights_stamp()
  assert e._weights_stamp() == s0
  k = m.weight_keys()[0]
  with torch.no_grad():
    m[k].add_(1.0)  # in place: the version counter
  s1 = e._weights_stamp()
  assert s1 != s0
  m[k] = m[k].clone()  # the entry replaced: the dict's mutation counter (the new tensor's version starts at 0 again)
  s2 = e._weights_stamp()
  assert s2 != s1 and s2[0] == s1[0] + 1
  for mutate in (lambda: m.update({k: m[k]}), lambda: m.setdefault('zz_probe', torch.zeros(1)), lambda: m.pop('zz_probe'),
                 lambda: m.__ior__({})):
    before = m._mut
    mutate()
    assert m._mut == before + 1
  plain = ra_engine.DecodeEngine(m.dims, dict(m))
  p0 = plain._weights_stamp()
  assert isinstance(p0, tuple) and len(p0) == len(m.weight_keys()) and plain._weights_stamp() == p0
  with torch.no_grad():
    m[k].add_(1.0)
  assert plain._weights_stamp() != p0


def test_global_pack_cache_keeps_only_the_latest_version_of_a_weight():
  """the train-mode nnlib closures pack their filters into ra_train._PACK, whose keys carry the weight's version; an
  in-place optimizer step used to leave one dead set of packs per step behind.  A newer version of the same (tensor, geometry)
  replaces the entry, other geometries of the tensor stay, and the cache is capped."""
  import ra_train as rt
  c = rt._LatestVersionPack()
  c[(100, 0, 3, 8, 4, 0, False)] = ('w', 'p0')
  c[(100, 1, 3, 8, 4, 0, False)] = ('w', 'p1')     # the weight moved in place: replaces p0
  c[(100, 1, 3, 8, 8, 0, False)] = ('w', 'q1')     # another geometry of the same weight: kept beside it
  c[('shift', 200, 0, 16)] = ('b', 's0')
  c[('shift', 200, 5, 16)] = ('b', 's5')
  c[('split', 100, 1, True)] = ('w', 1)
  c[('split', 100, 2, True)] = ('w', 2)
  assert sorted(map(str, c)) == sorted(map(str, [(100, 1, 3, 8, 4, 0, False), (100, 1, 3, 8, 8, 0, False), ('shift', 200, 5, 16),
                                                 ('split', 100, 2, True)]))
  assert c.get((100, 0, 3, 8, 4, 0, False)) is None and c[(100, 1, 3, 8, 4, 0, False)][1] == 'p1'
  for i in range(c.CAP + 500):
    c[(1000 + i, 0, 1)] = i
  assert len(c) == c.CAP and len(c._latest) == c.CAP
  c.clear()
  assert not c and not c._latest
  assert isinstance(rt._PACK, rt._LatestVersionPack)




def test_pipeline_end_of_stream_rule(monkeypatch):
  """DecodePipeline._ends_soon (round 6): batches stop waiting for company only when the caller has said how many follow and
  everything left fits the slots one batch each."""
  import full_model
  monkeypatch.setenv('RA_PIPE_ENDGAME', '1')  # (off by default: measured slower at cfg2, profiles/r06_pipeline_endgame.txt)

  class M:
    OUTPUTS = ()
  p = full_model.DecodePipeline(M(), depth=4, coalesce=2)
  assert not p._ends_soon(None) and not p._ends_soon(4) and p._ends_soon(3) and p._ends_soon(0)
  p.group = [dict(B=8)]
  assert not p._ends_soon(3) and p._ends_soon(2)
  assert not p.full(remaining=None) and not p.full(remaining=5)  # joining a waiting group needs no slot
  p.free = []
  assert p.full(remaining=2)  # ... but going out alone does
  assert not full_model.DecodePipeline(M(), depth=4, coalesce=1)._ends_soon(0)
  monkeypatch.setenv('RA_PIPE_ENDGAME', '0')
  assert not full_model.DecodePipeline(M(), depth=4, coalesce=2)._ends_soon(0)
