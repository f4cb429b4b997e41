"""Parity of the evaluation post-processing and metrics (csrc/ra_eval.hip through the C ABI;
host modules utils/postprocess.py and analysis.py with the reference's names) against the
NumPy oracle restating utils/postprocess.py:5-147 and analysis.py:314-787.
Binary masks and counts are exact (integers below 2^24 in float32); ratios within 1e-6."""
import numpy as np
import pytest
import torch

import ra_oracle as ora
from test_loss_gpu import dev, synth_gt

pytestmark = pytest.mark.gpu


def noisy_prediction(rng, y_gt, s_gt):
  """Soft masks that roughly follow the ground truth: shifted / eroded instances in a shuffled
  order, one spurious instance, one missed instance per image, plus low-level noise."""
  B, T, H, W = y_gt.shape
  y = 0.05 * rng.rand(B, T, H, W).astype(np.float32)
  s = 0.1 * rng.rand(B, T).astype(np.float32)
  for b in range(B):
    k = int(s_gt[b].sum())
    order = rng.permutation(T)
    for j in range(max(k - 1, 0)):  # the last ground-truth instance is missed
      m = np.roll(y_gt[b, j], (rng.randint(-3, 4), rng.randint(-3, 4)), axis=(0, 1))
      y[b, order[j]] = np.maximum(y[b, order[j]], m * rng.uniform(0.6, 1.0))
      s[b, order[j]] = rng.uniform(0.6, 1.0)
    if k < T:  # a spurious detection
      yy, xx = np.mgrid[0:H, 0:W]
      y[b, order[k]] = np.maximum(y[b, order[k]], 0.9 * (((yy - 5) ** 2 + (xx - 5) ** 2) < 16))
      s[b, order[k]] = 0.8
  return y, s


@pytest.mark.parametrize('B,T,H,W,thresh,use_fg,tiny', [(2, 6, 32, 32, 0.5, False, 0), (3, 16, 64, 48, 0.3, True, 30),
                                                        (1, 21, 40, 40, 0.6, True, 0), (2, 5, 128, 128, 0.5, False, 0)])
def test_postprocess(cuda, B, T, H, W, thresh, use_fg, tiny):
  from utils import postprocess as pp
  rng = np.random.RandomState(T * 3 + H)
  y_gt, s_gt = synth_gt(rng, B, T, H, W, max_inst=max(2, T - 2))
  y, s = noisy_prediction(rng, y_gt, s_gt)
  y[0, :, 3, 3] = 0.7  # an exact tie across all instances: numpy.argmax keeps the first
  s[0, :] = np.maximum(s[0, :], 0.75)
  s[0, :] = s[0, 0]
  fg = (y_gt.max(axis=1) > 0).astype(np.float32) if use_fg else None
  ref_y, ref_s = ora.postprocess(y.astype(np.float64), s.astype(np.float64), thresh, fg=fg, remove_tiny_threshold=tiny)
  got_y, got_s, uni = pp.postprocess(dev(y, cuda), dev(s, cuda), thresh, fg=None if fg is None else dev(fg, cuda),
                                     remove_tiny_threshold=tiny)
  torch.cuda.synchronize()
  assert (got_y.cpu().numpy() == ref_y).all()
  assert (got_s.cpu().numpy() == ref_s).all()
  if uni is not None:
    assert (uni.cpu().numpy() == ref_y.max(axis=1)).all()
  # the individual operators with the reference's names
  dy, ds = dev(y, cuda), dev(s, cuda)
  yc, sh = pp.apply_confidence(dy, ds)
  r_yc, r_sh = ora.pp_apply_confidence(y.astype(np.float64), s.astype(np.float64))
  assert np.abs(yc.cpu().numpy() - r_yc).max() < 1e-6 and (sh.cpu().numpy() == r_sh).all()
  one = pp.apply_one_label(yc)
  assert np.abs(one.cpu().numpy() - ora.pp_apply_one_label(yc.cpu().numpy().astype(np.float64))).max() == 0
  thr = pp.apply_threshold(one, thresh)
  assert (thr.cpu().numpy() == ora.pp_apply_threshold(one.cpu().numpy(), thresh)).all()
  rt_y, rt_s = pp.remove_tiny(thr, sh, threshold=25)
  o_y, o_s = ora.pp_remove_tiny(thr.cpu().numpy().astype(np.float64), sh.cpu().numpy().astype(np.float64), 25)
  assert (rt_y.cpu().numpy() == o_y).all() and (rt_s.cpu().numpy() == o_s).all()
  if fg is not None:
    assert (pp.mask_foreground(thr, dev(fg, cuda)).cpu().numpy() == thr.cpu().numpy() * fg[:, None]).all()
  # the evaluator's cv2 steps (postprocess.py:55-106; round 5): 5 x 5 dilation — exact — and linear resize + bilateral filter
  mo = pp.morph(thr)
  assert (mo.cpu().numpy() == ora.pp_morph(thr.cpu().numpy().astype(np.float64))).all()
  soft = yc[:, :, ::2, ::2].contiguous()
  up = pp.upsample(soft, thr)
  assert tuple(up.shape) == tuple(thr.shape)
  assert np.abs(up.cpu().numpy() - ora.pp_upsample(soft.cpu().numpy(), thr.shape[2], thr.shape[3])).max() < 2e-6


@pytest.mark.parametrize('B,T,H,W', [(3, 8, 64, 64), (2, 16, 48, 80), (2, 21, 32, 32)])
def test_eval_metrics(cuda, B, T, H, W):
  import analysis
  rng = np.random.RandomState(B + T + H)
  y_gt, s_gt = synth_gt(rng, B, T, H, W, max_inst=T - 2)
  y, s = noisy_prediction(rng, y_gt, s_gt)
  y_gt[B - 1] = 0.0   # an image without ground truth (num_obj is clamped to 1)
  s_gt[B - 1] = 0.0
  y_bin, s_hard = ora.postprocess(y.astype(np.float64), s.astype(np.float64), 0.5)
  if B > 2:
    y_bin[1] = 0.0    # an image without detections
  ref = ora.eval_metrics(y_bin, y_gt.astype(np.float64), s_gt.astype(np.float64))
  results = {'y_out': dev(y_bin, cuda), 'y_gt': dev(y_gt, cuda), 's_out': dev(s_hard, cuda), 's_gt': dev(s_gt, cuda)}
  for name in ('sbd', 'wt_cov', 'unwt_cov', 'fg_iou', 'fg_dice', 'avg_fp', 'avg_fn', 'count_acc', 'count_mse',
               'dic', 'dic_abs', 'avg_pr', 'avg_re', 'obj_pr', 'obj_re'):
    got = analysis.create_analyzer(name)(results).cpu().numpy()
    assert got.shape == ref[name].shape, name
    assert np.abs(got - ref[name]).max(initial=0.0) < 1e-6, name
  assert np.abs(results['iou_pairwise'].cpu().numpy() - ref['iou_pairwise']).max() < 1e-6
  assert np.abs(analysis.f_iou_pairwise(results['y_out'], results['y_gt']).cpu().numpy() - ref['iou_pairwise']).max() < 1e-6
  assert (analysis.f_count_out(results['y_out']).cpu().numpy() == (y_bin.sum(axis=(2, 3)) > 0)).all()
  with pytest.raises(Exception):
    analysis.create_analyzer('no_such_metric')


def test_decode_postprocess_metrics_end_to_end(cuda):
  """full_model_eval's chain on a decoded batch: decode -> post-process -> metrics, against the
  oracle's decode -> post-process -> metrics (cfg1-sized CVPPP model, seeded weights)."""
  import analysis
  import full_model
  from utils import postprocess as pp
  opt = ora.make_opt('cvppp', 128, 128, 5)
  P = ora.random_params(opt, 16)
  rng = np.random.RandomState(23)
  x = rng.rand(2, 128, 128, 3).astype(np.float32)
  y_gt, s_gt = synth_gt(rng, 2, 5, 128, 128, max_inst=4)
  fwd = ora.full_model_forward(opt, P, x)
  ref_y, ref_s = ora.postprocess(fwd['y_out'], fwd['s_out'], 0.5)
  ref = ora.eval_metrics(ref_y, y_gt.astype(np.float64), s_gt.astype(np.float64))
  m = full_model.get_model(opt).load_weights(P)
  y_out, s_out = m.run(['y_out', 's_out'], {'x': x, 'phase_train': False})
  y_bin, s_hard, _ = pp.postprocess(y_out, s_out, 0.5)
  torch.cuda.synchronize()
  # masks within 1e-3 of the oracle can flip pixels that sit on the threshold: allow a handful
  flips = (y_bin.cpu().numpy() != ref_y).sum()
  assert flips <= 20
  results = {'y_out': y_bin, 'y_gt': dev(y_gt, cuda), 's_out': s_hard, 's_gt': dev(s_gt, cuda)}
  for name in ('sbd', 'wt_cov', 'unwt_cov', 'fg_dice', 'dic'):
    got = analysis.create_analyzer(name)(results).cpu().numpy()
    assert np.abs(got - ref[name]).max() < 5e-3, name


def test_eval_driver_writes_predictions_and_metrics(cuda, tmp_path):
  """full_model_train.py --init_only -> full_model_eval.py with ground truth in the input: the
  reference's write_log chain (full_model_eval.py:97-139) end to end through the CLI surface."""
  import yaml
  import analysis
  import full_model
  import full_model_eval
  import full_model_train
  from utils import postprocess as pp
  res = str(tmp_path / 'results')
  with pytest.raises(Exception):  # no controller input selected: a clear error, not a kernel failure
    full_model_train.main(['--init_only', '--results', res, '--model_id', 'bad', '--inp_height', '64',
                           '--inp_width', '64', '--timespan', '4'])
  full_model_train.main(['--init_only', '--results', res, '--model_id', 'm0', '--inp_height', '64',
                         '--inp_width', '64', '--timespan', '4', '--ctrl_add_inp', '--ctrl_add_canvas',
                         '--attn_add_inp', '--attn_add_canvas', '--fixed_gamma',  # run_cvppp.sh:44-72
                         '--ctrl_cnn_filter_size', '3,3,3,3', '--ctrl_cnn_depth', '8,8,16,16',
                         '--ctrl_cnn_pool', '1,2,1,2', '--attn_cnn_filter_size', '3,3,3,3,3,3',
                         '--attn_cnn_depth', '8,8,16,16,32,32', '--attn_cnn_pool', '1,2,1,2,1,2',
                         '--attn_dcnn_filter_size', '3,3,3,3,3,3,3', '--attn_dcnn_depth', '32,32,16,16,8,8,1',
                         '--attn_dcnn_pool', '2,1,2,1,2,1,1'])
  rng = np.random.RandomState(4)
  x = rng.rand(3, 64, 64, 3).astype(np.float32)
  y_gt, s_gt = synth_gt(rng, 3, 4, 64, 64, max_inst=3)
  inp = str(tmp_path / 'in.npz')
  np.savez(inp, x=x, y_gt=y_gt, s_gt=s_gt)
  full_model_eval.main(['--model_id', 'm0', '--results', res, '--input', inp, '--batch_size', '2',
                        '--threshold_list', '0.3,0.5', '--analyzers', 'sbd,wt_cov,dic,avg_pr'])
  out_dir = tmp_path / 'results' / 'm0' / 'output_valid'
  pred = np.load(str(out_dir / 'pred_rank0.npz'))
  assert pred['y_out'].shape == (3, 4, 64, 64) and pred['s_out'].shape == (3, 4)
  summary = yaml.safe_load(open(str(out_dir / 'metrics_rank0.yaml')))
  assert sorted(summary) == ['0.30', '0.50'] and sorted(summary['0.50']) == ['avg_pr', 'dic', 'sbd', 'wt_cov']
  # the same numbers computed directly
  with open(str(tmp_path / 'results' / 'm0' / 'model_opt.yaml')) as f:
    opt = yaml.safe_load(f)
  m = full_model.get_model(opt).load_weights(dict(np.load(str(tmp_path / 'results' / 'm0' / 'weights.npz'))))
  y, s = m.run(['y_out', 's_out'], {'x': x, 'phase_train': False})
  # the reference's chain (full_model_eval.py:112-124): pp.upsample — resize + bilateral filter — runs even at equal size
  yc, s_hard = pp.apply_confidence(y, s)
  y_bin = pp.apply_threshold(pp.apply_one_label(pp.upsample(yc, dev(y_gt, cuda))), 0.5)
  r = {'y_out': y_bin, 'y_gt': dev(y_gt, cuda), 's_out': s_hard, 's_gt': dev(s_gt, cuda)}
  assert abs(summary['0.50']['sbd']['mean'] - float(analysis.f_symmetric_best_dice(r).mean())) < 1e-6
  assert summary['0.50']['dic']['count'] == 3
  # --fused_postprocess (not a reference flag): no bilateral step at equal size, one fused pass
  full_model_eval.main(['--model_id', 'm0', '--results', res, '--input', inp, '--batch_size', '2', '--fused_postprocess',
                        '--threshold_list', '0.5', '--analyzers', 'sbd'])
  summary = yaml.safe_load(open(str(out_dir / 'metrics_rank0.yaml')))
  y_bin, s_hard, _ = pp.postprocess(y, s, 0.5)
  r = {'y_out': y_bin, 'y_gt': dev(y_gt, cuda), 's_out': s_hard, 's_gt': dev(s_gt, cuda)}
  assert abs(summary['0.50']['sbd']['mean'] - float(analysis.f_symmetric_best_dice(r).mean())) < 1e-6


@pytest.mark.parametrize('H,W,C,pad', [(32, 32, 3, 8), (24, 40, 8, 5), (16, 16, 1, 0)])
def test_random_transformation(cuda, H, W, C, pad):
  """image_ops.random_transformation: every combination of the draws against the oracle, the
  identity at evaluation, and the reference-named wrapper applying the same draws to x and y."""
  import image_ops
  import ra_ops as ops
  rng = np.random.RandomState(H + C)
  x = rng.rand(2, H, W, C).astype(np.float32)
  for off_y, off_x in ((0, 0), (pad, pad), (2 * pad, max(2 * pad - 1, 0))):
    for fv in (False, True):
      for fh in (False, True):
        for tr in ((False, True) if H == W else (False,)):
          got = ops.random_transform(dev(x, cuda), pad, off_y, off_x, fv, fh, tr).cpu().numpy()
          assert (got == ora.random_transformation(x, pad, off_y, off_x, fv, fh, tr)).all()
  xd = dev(x[..., :3] if C >= 3 else np.repeat(x, 3, axis=3), cuda)
  y = dev(rng.rand(2, 4, H, W), cuda)
  same = image_ops.random_transformation(xd, pad, False, y=y)
  assert same['x'] is xd and same['y'] is y
  g = torch.Generator().manual_seed(5)
  r = image_ops.random_transformation(xd, pad, True, rnd_transpose=(H == W), y=y, generator=g)
  kw = r['_draws']
  assert (r['x'].cpu().numpy() == ora.random_transformation(xd.cpu().numpy(), **kw)).all()
  yr = ora.random_transformation(y.cpu().numpy().reshape(8, H, W), **kw).reshape(2, 4, H, W)
  assert (r['y'].cpu().numpy() == yr).all()
  if C == 3:  # the colour jitter (image_ops.py:99-103) on the cropped / flipped image, for given and for drawn factors
    col = dict(hue=0.07, saturation=1.08, brightness=-0.06, contrast=0.93)
    r2 = image_ops.random_transformation(xd, pad, True, rnd_transpose=False, rnd_colour=True, y=y, draws=dict(kw, **col))
    ref = ora.colour_jitter(ora.random_transformation(xd.cpu().numpy(), **dict(kw, transpose=kw['transpose'])), **col)
    assert np.abs(r2['x'].cpu().numpy() - ref).max() < 2e-6 and (r2['y'].cpu().numpy() == yr).all()
    r3 = image_ops.random_transformation(xd, pad, True, rnd_transpose=False, rnd_colour=True, generator=torch.Generator().manual_seed(9))
    d3 = r3['_draws']
    assert -0.1 <= d3['hue'] <= 0.1 and 0.9 <= d3['saturation'] <= 1.1 and -0.1 <= d3['brightness'] <= 0.1 and 0.9 <= d3['contrast'] <= 1.1
    ref3 = ora.colour_jitter(ora.random_transformation(xd.cpu().numpy(), **{k: d3[k] for k in kw}),
                             **{k: d3[k] for k in ('hue', 'saturation', 'brightness', 'contrast')})
    assert np.abs(r3['x'].cpu().numpy() - ref3).max() < 2e-6
