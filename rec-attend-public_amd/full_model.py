"""`full_model.get_model` of the reference (full_model.py:13-1099) on MI355X kernels.

get_model(opt, is_training=True) takes the reference's `model_opt` dict
(full_model_train.py:581-658) and returns a dict-like `Model`:
  * every weight registered under the reference's key (`ctrl_cnn_w_0`,
    `ctrl_cnn_0_3_beta`, `ctrl_lstm_w_xi`, `glimpse_mlp_w_0`, `ctrl_mlp_w_0`,
    `attn_cnn_*`, `attn_dcnn_*`, `score_mlp_w_0`; nnlib.py:120-127,206-211,333-335,
    471-474,611-623) as a float32 device tensor;
  * `model.run(outputs, feed)` — the counterpart of the harness contract
    `sess.run([model[k] for k in outputs], feed_dict)` (runner.py:91-105): feed keys `x`
    [B,H,W,3], `phase_train`, optional `d_in` [B,H,W,8], `y_in` [B,H,W,nc]; outputs are the
    reference's output keys (full_model.py:853-907): `y_out` [B,T,H,W], `s_out` [B,T],
    `x_patch`, `y_out_patch`, `attn_box`, `attn_ctr`, `attn_size`, `attn_top_left`,
    `attn_bot_right`, `attn_ctr_norm`, `attn_lg_size`, `ctrl_rnn_glimpse_map`.
The forward is the eval graph (phase_train=False; use_knob does not act at eval,
full_model.py:744-773,826-841).  With `y_gt` / `s_gt` in the feed the loss / statistics head
of the training graph (full_model.py:913-1097) is available as outputs too (`loss`, `iou_soft`,
`match`, ... — Model.LOSS_OUTPUTS; csrc/ra_loss.hip).  `train_step` (backward + Adam,
full_model.py:1039-1057) lives in ra_train.py.
"""
import os
import numpy as np
import torch

import modellib
import nnlib as nn
import ra_engine
import ra_ops as ops
from ra_native import RecAttendError


def _get(opt, key, default):
  return opt[key] if key in opt else default


def derive_dims(opt, box_model=False):
  """Shape / flag bookkeeping of the graph builder (full_model.py:18-160,239-258,305-313,
  455-459,494-502; box_model.py:16-82,343-352)."""
  d = {'T': opt['timespan'], 'H': opt['inp_height'], 'W': opt['inp_width'],
       'D': opt['inp_depth'], 'Fh': opt['filter_height'], 'Fw': opt['filter_width']}
  if d['H'] % 4 or d['W'] % 4:
    raise RecAttendError('inp_height / inp_width must be multiples of 4')
  if 'add_d_out' in opt:
    add_d, add_y = bool(opt['add_d_out']), bool(opt['add_y_out'])
  else:
    add_d = add_y = False
  assert (add_d and add_y) or (not add_d and not add_y)  # full_model.py:206
  nsc = _get(opt, 'num_semantic_classes', 1)
  d.update(add_d_out=add_d, add_y_out=add_y, nsc=nsc)
  if box_model:
    ctrl = (True, True, add_d, add_y)
    attn = ctrl
  else:
    if 'attn_add_d_out' in opt:
      attn = (opt['attn_add_inp'], opt['attn_add_canvas'], opt['attn_add_d_out'],
              opt['attn_add_y_out'])
    else:
      attn = (True, True, add_d, add_y)
    if 'ctrl_add_d_out' in opt:
      ctrl = (opt['ctrl_add_inp'], opt['ctrl_add_canvas'], opt['ctrl_add_d_out'],
              opt['ctrl_add_y_out'])
    else:
      ctrl = (not add_d, not add_d, add_d, add_y)
  ctrl, attn = tuple(bool(v) for v in ctrl), tuple(bool(v) for v in attn)
  depth = lambda fl: (d['D'] if fl[0] else 0) + (1 if fl[1] else 0) + (8 if fl[2] else 0) + (
      nsc if fl[3] else 0)
  d['ctrl_in'], d['attn_in'] = ctrl, attn
  d['C0p'] = -(-(d['D'] + 1 + (8 if add_d else 0) + (nsc if add_y else 0)) // 4) * 4
  d['ccnn_nlayers'] = len(opt['ctrl_cnn_filter_size'])
  d['ccnn_filters'] = list(opt['ctrl_cnn_filter_size'])
  d['ccnn_channels'] = [depth(ctrl)] + list(opt['ctrl_cnn_depth'])
  if d['ccnn_channels'][0] == 0:
    raise RecAttendError('the controller CNN has no input: pass at least one of ctrl_add_inp / '
                         'ctrl_add_canvas / ctrl_add_d_out / ctrl_add_y_out (full_model.py:140-149; '
                         'the run scripts pass --ctrl_add_inp --ctrl_add_canvas)')
  d['ccnn_pool'] = list(opt['ctrl_cnn_pool'])
  sub = int(np.prod(d['ccnn_pool']))
  d['gh'], d['gw'] = d['H'] // sub, d['W'] // sub
  d['G'] = d['gh'] * d['gw']
  d['hid'] = opt['ctrl_rnn_hid_dim']
  d['iters'] = opt['num_ctrl_rnn_iter']
  d['n_gmlp'] = opt['num_glimpse_mlp_layers']
  d['n_cmlp'] = opt['num_ctrl_mlp_layers']
  d['mlp_dim'] = opt['ctrl_mlp_dim']
  d['squash'] = bool(opt['squash_ctrl_params'])
  d['fixed_var'] = bool(_get(opt, 'fixed_var', True if box_model else False))
  d['dynamic_var'] = bool(_get(opt, 'dynamic_var', False))
  d['use_bn'] = bool(opt['use_bn'])
  d['attn_box_padding_ratio'] = opt['attn_box_padding_ratio']
  if box_model:
    d['fixed_gamma'] = True
    return d
  d['fixed_gamma'] = bool(opt['fixed_gamma'])
  d['disable_overwrite'] = bool(_get(opt, 'disable_overwrite', True))
  d['acnn_nlayers'] = len(opt['attn_cnn_filter_size'])
  d['acnn_filters'] = list(opt['attn_cnn_filter_size'])
  d['acnn_channels'] = [depth(attn)] + list(opt['attn_cnn_depth'])
  d['acnn_pool'] = list(opt['attn_cnn_pool'])
  asub = int(np.prod(d['acnn_pool']))
  d['core_depth'] = d['acnn_channels'][-1]
  d['core_dim'] = (d['Fh'] // asub) * (d['Fw'] // asub) * d['core_depth']
  d['adcnn_nlayers'] = len(opt['attn_dcnn_filter_size'])
  d['adcnn_filters'] = list(opt['attn_dcnn_filter_size'])
  d['adcnn_unpool'] = list(opt['attn_dcnn_pool'])
  d['adcnn_channels'] = [d['core_depth']] + list(opt['attn_dcnn_depth'])
  add_skip = bool(_get(opt, 'add_skip_conn', True))
  d['add_skip_conn'] = add_skip
  skip_flags = _get(opt, 'attn_cnn_skip', [add_skip] * d['acnn_nlayers'])
  # NB: the reference's CLI leaves the raw flag string here ('1,0,1,..'), whose characters are
  # all truthy; iterating it reproduces "all skips on".
  skip_rev = list(skip_flags[::-1])
  if add_skip:
    ch_rev = d['acnn_channels'][::-1][1:] + [d['acnn_channels'][0]]
    d['skip_ch'] = [0] + [ch if sk else 0 for sk, ch in zip(skip_rev, ch_rev)]
    d['skip_ch'] = (d['skip_ch'] + [0] * d['adcnn_nlayers'])[:d['adcnn_nlayers']]
  else:
    d['skip_ch'] = None
  return d


class Model(dict):
  """The reference's `model` dict plus run().  Tensor-valued entries are weights."""

  OUTPUTS = ('y_out', 's_out', 'x_patch', 'y_out_patch', 'attn_box', 'attn_ctr', 'attn_size',
             'attn_top_left', 'attn_bot_right', 'attn_ctr_norm', 'attn_lg_size',
             'ctrl_rnn_glimpse_map', 'ctrl_out', 'h_core', 'canvas')
  # the loss / statistics head of the training graph, forward only (full_model.py:913-1097):
  # needs y_gt and s_gt in the feed
  LOSS_OUTPUTS = ops.STAT_NAMES + ('match', 'match_box', 'attn_box_gt', 'attn_top_left_gt',
                                   'attn_bot_right_gt')
  TRAIN_ONLY = ('train_step', 'learn_rate', 'gt_knob_prob_box', 'gt_knob_prob_segm')

  def __init__(self, opt, dims, box_model=False):
    dict.__init__(self)
    self._mut = 0
    self.opt = dict(opt)
    self.dims = dims
    self.box_model = box_model
    self.engine = None

  # every change of the dict itself is counted (`_mut`): the decode engine re-packs its weights when an entry is replaced or a
  # tensor is modified in place, and tells the two apart cheaply — this counter, and the sum of the tensors' version counters
  def __setitem__(self, k, v):
    self._mut += 1
    dict.__setitem__(self, k, v)

  def __delitem__(self, k):
    self._mut += 1
    dict.__delitem__(self, k)

  def update(self, *a, **kw):
    self._mut += 1
    dict.update(self, *a, **kw)

  def setdefault(self, k, default=None):
    self._mut += 1
    return dict.setdefault(self, k, default)

  def pop(self, *a):
    self._mut += 1
    return dict.pop(self, *a)

  def popitem(self):
    self._mut += 1
    return dict.popitem(self)

  def clear(self):
    self._mut += 1
    dict.clear(self)

  def __ior__(self, other):
    self._mut += 1
    dict.update(self, other)
    return self

  def weight_keys(self):
    return sorted(k for k, v in self.items() if isinstance(v, torch.Tensor))

  def load_weights(self, weights, strict=True):
    """weights: mapping name -> array (e.g. an .npz of the reference's weights.h5 keys,
    full_model_read.py:33-71, plus the BN EMA statistics).

    strict (default): every registered tensor must be present in `weights` — in particular the
    `*_ema_mean` / `*_ema_var` BatchNorm statistics, which the reference's weights.h5 export
    omits (they live only in the TF checkpoint, nnlib.py:121-127): left at their initial zeros
    every BN layer would compute gamma * x / sqrt(1e-3).  Keys of `weights` the model does not
    register are reported with a warning.  strict=False loads what matches and returns quietly
    (the pre-training hand-off of a sub-network, full_model.py:271-284)."""
    known = set(self.weight_keys())
    given = set(k for k in weights.keys() if not k.startswith('optim/'))  # optimizer state of a training checkpoint: ra_train's
    if strict:
      missing = sorted(known - given)
      if missing:
        raise RecAttendError(
            'load_weights: %d registered tensors are missing from the archive (first: %s); BN EMA '
            'statistics are not part of the reference weights.h5 export — add them or pass '
            'strict=False' % (len(missing), ', '.join(missing[:4])))
    extra = sorted(given - known)
    if extra:
      import warnings
      warnings.warn('load_weights: ignoring %d unknown keys (first: %s)' % (len(extra), ', '.join(extra[:4])))
    for k, v in weights.items():
      if k not in known:
        continue
      v = torch.as_tensor(np.asarray(v, dtype=np.float32))
      if tuple(v.shape) != tuple(self[k].shape):
        raise RecAttendError('weight %s: shape %r != %r' % (k, tuple(v.shape),
                                                           tuple(self[k].shape)))
      self[k].copy_(v)
    return self

  def state_dict_numpy(self):
    return {k: self[k].detach().cpu().numpy() for k in self.weight_keys()}

  def run(self, outputs, feed, as_numpy=False):
    single = isinstance(outputs, str)
    names = [outputs] if single else list(outputs)
    for n in names:
      if n not in self.OUTPUTS and n not in self.LOSS_OUTPUTS and n not in self.TRAIN_ONLY:
        raise KeyError(n)
    if 'train_step' in names or nn._is_train(feed.get('phase_train', False)):
      return self._run_train(names, feed, single, as_numpy)
    if any(n in self.TRAIN_ONLY for n in names):
      raise RecAttendError('outputs %r need phase_train = True' % [n for n in names if n in self.TRAIN_ONLY])
    d = self.dims
    want_loss = any(n in self.LOSS_OUTPUTS for n in names)
    if want_loss and self.box_model:
      raise NotImplementedError('the loss head is built for full_model only')
    b = self.engine.forward(feed['x'], d_in=feed.get('d_in'), y_in=feed.get('y_in'),
                            y_gt=feed.get('y_gt') if self.box_model else None,
                            noise=feed.get('noise'),
                            want_box=want_loss or 'attn_box' in names)
    b.check_status()
    head = self._loss_head(b, feed) if want_loss else {}
    res = [head[n] if n in head else self._fetch(n, b) for n in names]
    if as_numpy:
      torch.cuda.synchronize()
      res = [r.detach().cpu().numpy() for r in res]
    return res[0] if single else res

  def pipeline(self, depth=4, max_images=None, co_resident=None, streams=None, coalesce=1):
    """`depth` slots of this model decoding concurrently (DecodePipeline below), `coalesce` submitted batches per slot."""
    return DecodePipeline(self, depth, max_images, co_resident, streams, coalesce)

  def _run_train(self, names, feed, single, as_numpy):
    """sess.run([loss, train_step], feed{x, y_gt, s_gt, phase_train=True}) (full_model_train.py:107):
    the training graph on BatchNorm batch statistics; fetching `train_step` applies one optimizer
    step (ra_train.TrainStep: backward, gradient all-reduce over the ranks, clip + Adam, EMA)."""
    import ra_train
    if 'y_gt' not in feed or 's_gt' not in feed:
      raise RecAttendError('the training graph needs y_gt and s_gt in the feed')
    if getattr(self, 'trainer', None) is None:
      self.trainer = (ra_train.BoxTrainStep if self.box_model else ra_train.TrainStep)(self)
    tr = self.trainer
    if 'noise' in feed and 'knobs' not in feed:  # box_model's canvas noise (box_model.py:500-502), as at eval
      feed = dict(feed, knobs={'noise': feed['noise']})
    if 'train_step' in names:
      extra = {k: feed[k] for k in ('d_in', 'y_in') if feed.get(k) is not None}
      out = tr.run(feed['x'], feed['y_gt'], feed['s_gt'], knobs=feed.get('knobs'), generator=feed.get('generator'),
                   aug=feed.get('aug'), **extra)
    else:
      with torch.no_grad():
        extra = {k: feed[k] for k in ('d_in', 'y_in') if feed.get(k) is not None}
        _, out, _ = tr.forward_loss(feed['x'], feed['y_gt'], feed['s_gt'], knobs=feed.get('knobs'),
                                    generator=feed.get('generator'), **extra)
      out = dict(out)
      out['learn_rate'] = ra_train.learn_rate(self.opt, tr.bucket.global_step)
    res = []
    for n in names:
      if n == 'train_step':
        res.append(None)
      elif n == 'learn_rate':
        res.append(out['learn_rate'])
      elif n in out:
        res.append(out[n])
      else:
        raise NotImplementedError('output %r is not available from the training graph' % n)
    if as_numpy:
      torch.cuda.synchronize()
      tr.flush_status()  # numpy outputs = the synchronous sess.run: this step's solver statuses are checked now
      res = [r.detach().cpu().numpy() if isinstance(r, torch.Tensor) else r for r in res]
    return res[0] if single else res

  def _loss_head(self, eng, feed):
    """full_model.py:913-1097 on the decoded batch (phase_train False: the GT knobs are off and
    the per-timestep iou_soft_box of the use_knob branch, :756-758, equals the pairwise f_iou)."""
    opt = self.opt
    if 'y_gt' not in feed or 's_gt' not in feed:
      raise RecAttendError('loss / statistics outputs need y_gt and s_gt in the feed')
    if opt.get('box_loss_fn', 'iou') != 'iou' or opt.get('segm_loss_fn', 'iou') not in ('iou', 'wt_cov'):
      raise NotImplementedError('box_loss_fn %r / segm_loss_fn %r (built: iou; iou, wt_cov)' %
                                (opt.get('box_loss_fn'), opt.get('segm_loss_fn')))
    dev = eng.fetch('y_out').device
    as_t = lambda v: torch.as_tensor(np.asarray(v, dtype=np.float32)).to(dev) \
        if not isinstance(v, torch.Tensor) else v.to(device=dev, dtype=torch.float32)
    y_gt, s_gt = as_t(feed['y_gt']).contiguous(), as_t(feed['s_gt']).contiguous()
    y_out, s_out, attn_box = eng.fetch('y_out'), eng.fetch('s_out'), eng.fetch('attn_box')
    # get_gt_attn -> get_gt_box with min_padding = padding + 4 (full_model.py:561-566)
    params, box_gt = ops.gt_box(y_gt, float(opt['attn_box_padding_ratio']), float(opt['padding']) + 4.0)
    segm = ops.pair_stats(y_out, y_gt, want=('iou_soft', 'iou_hard', 'dice_hard', 'sum_b'))
    boxs = ops.pair_stats(attn_box, box_gt, want=('iou_soft',))
    fixed = bool(opt.get('fixed_order', False))
    match_real, st1 = ops.segm_match(segm['iou_soft'], s_gt)
    match_box, st2 = ops.segm_match(boxs['iou_soft'], s_gt)
    stats = ops.loss_stats(segm['iou_soft'], segm['iou_hard'], segm['dice_hard'], match_real,
                           boxs['iou_soft'], match_box, s_out, s_gt, segm['sum_b'], fixed_order=fixed,
                           segm_loss_fn=opt.get('segm_loss_fn', 'iou'),
                           loss_mix_ratio=float(opt.get('loss_mix_ratio', 1.0)))
    self.match_status = (st1, st2)
    ops.check_match_status(st1, 'f_segm_match(y_out, y_gt)')
    ops.check_match_status(st2, 'f_segm_match(attn_box, attn_box_gt)')
    head = {n: stats[i] for i, n in enumerate(ops.STAT_NAMES)}
    if fixed:
      ident = modellib.get_identity_match(s_gt.shape[0], s_gt.shape[1], s_gt)
      head['match'], head['match_box'] = ident, ident
    else:
      head['match'], head['match_box'] = match_real, match_box
    head['attn_box_gt'] = box_gt
    head['attn_top_left_gt'] = params[:, :, 0:2].contiguous()
    head['attn_bot_right_gt'] = params[:, :, 2:4].contiguous()
    return head

  def _fetch(self, name, eng):
    d = self.dims
    tb = lambda t: t.transpose(0, 1).contiguous()
    if name in ('y_out', 's_out', 'attn_box'):
      return eng.fetch(name).clone()
    b = {k: eng.fetch(k) for k in ('attn',)}
    if name in ('x_patch', 'y_out_patch', 'gmaps', 'ctrl_out', 'img'):
      b[name] = eng.fetch(name)
    if name == 'ctrl_rnn_glimpse_map':
      b['gmaps'] = eng.fetch('gmaps')
    if name == 'canvas':
      return eng.fetch('canvas').unsqueeze(-1).contiguous()
    a = b['attn']  # [T,B,16]
    if name == 'x_patch':
      xp = tb(b['x_patch'])
      sel = eng.attn_sel
      if sel != list(range(xp.shape[-1])):
        xp = xp[..., sel].contiguous()
      return xp
    if name == 'y_out_patch':
      return tb(b['y_out_patch'])
    if name == 'ctrl_rnn_glimpse_map':
      g = tb(b['gmaps'])
      return g.view(g.shape[0], d['T'], d['iters'], d['gh'], d['gw'])
    if name == 'ctrl_out':
      return tb(b['ctrl_out'])
    if name == 'attn_ctr':
      return tb(a[:, :, 0:2])
    if name == 'attn_size':
      return tb(a[:, :, 2:4])
    if name == 'attn_ctr_norm':
      return tb(a[:, :, 9:11])
    if name == 'attn_lg_size':
      return tb(a[:, :, 11:13])
    if name == 'attn_top_left':
      return tb(a[:, :, 0:2] - a[:, :, 2:4] / 2.0)  # modellib.py:850-852
    if name == 'attn_bot_right':
      return tb(a[:, :, 0:2] + a[:, :, 2:4] / 2.0)
    if name == 'h_core':
      raise KeyError('h_core is only available per step; fetch x_patch / y_out_patch instead')
    if name == 'canvas':
      return b['img'][..., d['D']:d['D'] + 1].contiguous()
    raise KeyError(name)


class DecodePipeline(object):
  """Several eval batches of one model in flight on one GPU.

  The reference's evaluator decodes its batches one `sess.run` after the other
  (full_model_eval.py:286-344).  Within one batch the timestep loop is strictly serial and half of
  each timestep is latency-bound (16-workgroup controller, patch-sized convs), so one batch alone
  leaves the chip idle between the controller-CNN launches.  Batches are independent, so slot k
  of the pipeline owns a DecodeEngine — its own activation buffers and its own linear HIP graph —
  and replays it on its own HIP stream: the streams land on different hardware queues and the
  latency-bound tail of one batch runs under the MFMA-bound controller CNN of the others.
  Measured at cfg2 on MI355X: 4.85 ms per batch alone, 2.50 ms per batch with 4 in flight, 2.48 ms
  (51.7k instance-timesteps/s) with 8 in flight on 4 streams; more than 4 streams lose again.  The slots decode with the one-workgroup-per-image
  controller unless depth x images x 16 workgroups fit the chip: the 16-workgroup controller spin-waits
  on its peers, and several such launches from different queues can each end up partially resident
  and starve one another (seen at cfg3 with 8-12 parts in flight: 0.1-3.7 s per step).  The HIP runtime multiplexes its
  streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, shared with the null stream and the
  graph-capture stream): two slots on one queue serialise (3.9 ms), so ra_native raises the default
  to 8 before the runtime starts.
  Every batch's results are bit-identical to a lone `model.run` (tests/test_full_model_gpu.py).

    pipe = model.pipeline(8)
    for feed in batches:
      if pipe.full():
        consume(pipe.collect())
      pipe.submit(['y_out', 's_out'], feed)
    while len(pipe):
      consume(pipe.collect())
  """

  def __init__(self, model, depth=4, max_images=None, co_resident=None, streams=None, coalesce=1):
    """streams: HIP streams the slots are dealt onto, round robin (default min(depth, 4): the GPU runs four
    queues at a time); with depth > streams a slot's batch is queued behind another slot's on the same stream,
    so the stream never waits for the host to notice a finished batch and submit the next.
    co_resident: engines decoding at the same time on this GPU if other pipelines run beside this one
    (default: streams); max_images: a batch with more images is decoded as ceil(B / max_images) near-equal parts,
    each on its own slot, and collect() returns them concatenated (KITTI's batch of 16 as 2 x 8: the
    16-workgroup controller needs all of a launch's workgroups co-resident, <= 14 images).
    coalesce (round 5): that many consecutively submitted batches are decoded by ONE slot as one forward over their
    images (eval-mode images are independent: a batch's results do not depend on its company); collect() still hands
    back one batch at a time.  With two batches of 8 per slot the controller CNN's share of a slot's timestep doubles
    while its latency-bound tail barely grows, and the four streams stop meeting in their tails with nothing for the
    matrix pipe to do: 53k -> 62k instance-timesteps/s at cfg2 with the same 64 images in flight
    (profiles/r05_pipeline_coalesce.txt).  A batch waits in the group until it is full; collect() / retire() / drain()
    of a waiting batch launch the group as it is."""
    if depth < 1:
      raise ValueError('depth must be >= 1')
    if max_images is not None and max_images < 1:
      raise ValueError('max_images must be >= 1')
    if coalesce < 1:
      raise ValueError('coalesce must be >= 1')
    self.model, self.depth, self.max_images = model, int(depth), max_images
    self.streams = max(1, min(self.depth, int(streams or 4)))
    self.co_resident = int(co_resident or self.streams)
    self.coalesce = int(coalesce)
    self.slots = None
    self.free = list(range(self.depth))
    self.group = []    # batches waiting for company: tickets whose 'launch' is still None
    self.pending = []  # tickets in submission order: dict(names, single, feed, to_host, B, launch, lo, hi)

  def _make_slots(self):
    proto = self.model.engine
    self.slots = []
    # RA_PIPE_PRIO: probing aid — "a,b,c,d" stream priorities (0 normal, -1 high) of the slots' streams
    prios = [int(v) for v in os.environ.get('RA_PIPE_PRIO', '').split(',') if v.strip()]
    streams = [torch.cuda.Stream(priority=prios[k % len(prios)]) if prios else torch.cuda.Stream() for k in range(self.streams)]
    for k in range(self.depth):
      eng = ra_engine.DecodeEngine(proto.d, self.model, box_model=proto.box)
      for flag in ('fuse_pairs', 'fuse_patch_pairs', 'ctrl_split', 'fuse_score',
                   'cache_first', 'fill_cache_inline', 'nsub', 'use_graph', 'use_wino', 'wino_unfuse', 'pair_wino', 'prefill_ride',
                   'fuse_extract_conv0', 'use_split', 'split_first', 'split_patch', 'box_iou_rects', 'ctrl_batch_xcd'):
        setattr(eng, flag, getattr(proto, flag))
      eng.co_resident = self.co_resident
      # slots on the same stream never run at the same time; with other pipelines beside this one (co_resident > streams) the
      # company is not ours to place: K2b then keeps its agent-scope exchange
      eng.xcd_slot, eng.xcd_slots = k % self.streams, (self.streams if self.co_resident <= self.streams else 0)
      self.slots.append((eng, streams[k % self.streams]))

  def __len__(self):
    return len(self.pending)

  def parts(self, B):
    """How many slots a launch of B images takes."""
    return 1 if not self.max_images else max(1, -(-int(B) // int(self.max_images)))

  def _ends_soon(self, remaining):
    """The caller has told how many batches follow (`remaining`), and what is left — the waiting group, the batch at hand and
    those — fits the slots one batch each: near the end of a finite stream batches stop waiting for company, so that the last
    ones keep ALL streams busy instead of filling half the slots twice as full (20 batches through four slots of two: the
    last four go out as 4 x 1, not 2 x 2)."""
    return (remaining is not None and self.coalesce > 1 and len(self.group) + 1 + int(remaining) <= self.depth and
            os.environ.get('RA_PIPE_ENDGAME', '0') == '1')

  def full(self, B=None, remaining=None):
    """The next submit() (of a batch of B images; default: one that takes one slot) would have to launch, and there is no
    slot for it: collect() / retire() first.  A batch that only joins the waiting group needs no slot.  remaining: as submit()."""
    if len(self.group) + 1 < self.coalesce and not self._ends_soon(remaining):
      return False
    waiting = sum(t['B'] for t in self.group)
    return len(self.free) < self.parts(waiting + (B if B is not None else 1))

  def submit(self, outputs, feed, to_host=False, remaining=None):
    """Start decoding one batch (eval outputs only); returns immediately.  to_host: the outputs are also
    copied to pinned host memory on the slot's stream (asynchronously, under the other batches' compute —
    y_out of a cfg2 batch is 134 MB, 2.7 ms of PCIe); collect() then returns NumPy arrays without a
    further copy.  remaining (round 6, optional): how many more batches the caller will submit after this one — an evaluator
    knows; see _ends_soon()."""
    if not torch.cuda.is_available():
      raise RecAttendError('the decode loop needs an MI355X (HIP device); no CPU fallback')
    single = isinstance(outputs, str)
    names = [outputs] if single else list(outputs)
    for n in names:
      if n not in self.model.OUTPUTS:
        raise KeyError(n)
    if nn._is_train(feed.get('phase_train', False)):
      raise RecAttendError('DecodePipeline decodes eval batches; training steps go through model.run')
    B = int(feed['x'].shape[0])
    waiting = sum(t['B'] for t in self.group)
    nparts = self.parts(waiting + B)
    if nparts > self.depth:
      raise RecAttendError('a batch of %d images needs %d slots of <= %d images; the pipeline has %d' %
                           (waiting + B, nparts, self.max_images, self.depth))
    ends = self._ends_soon(remaining)
    launches = len(self.group) + 1 >= self.coalesce or ends
    if launches and nparts > len(self.free):
      raise RecAttendError('%d of %d slots busy: collect() a batch before the next submit()' %
                           (self.depth - len(self.free), self.depth))
    if self.group and (self.group[0]['names'] != names or bool(self.group[0]['to_host']) != bool(to_host) or
                       set(k for k, v in self.group[0]['feed'].items() if v is not None and k != 'phase_train') !=
                       set(k for k, v in feed.items() if v is not None and k != 'phase_train')):
      self._launch_group()  # company must want the same outputs from the same kind of feed
    t = dict(names=names, single=single, feed=feed, to_host=to_host, B=B, launch=None, lo=0, hi=B)
    self.group.append(t)
    self.pending.append(t)
    if len(self.group) >= self.coalesce or ends:
      self._launch_group()

  def _launch_group(self):
    """One forward over the images of the waiting batches, on the next free slot(s)."""
    if not self.group:
      return
    members, self.group = self.group, []
    if self.slots is None:
      self._make_slots()
    names, to_host = members[0]['names'], members[0]['to_host']
    feeds = [m['feed'] for m in members]
    as_dev = lambda v: v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    if len(members) == 1:
      feed = feeds[0]
    else:  # the engines copy their inputs into their own buffers anyway: one concatenation per fed tensor
      feed = {}
      for k in ('x', 'd_in', 'y_in', 'y_gt'):
        if feeds[0].get(k) is not None:
          feed[k] = torch.cat([as_dev(f[k]).to('cuda', non_blocking=True) for f in feeds], dim=0)
      if feeds[0].get('noise') is not None:  # box_model: [T, B, H, W]
        feed['noise'] = torch.cat([as_dev(f['noise']).to('cuda') for f in feeds], dim=1)
    B = sum(m['B'] for m in members)
    nparts = self.parts(B)
    if nparts > len(self.free):
      raise RecAttendError('%d of %d slots busy: collect() a batch before the next submit()' %
                           (self.depth - len(self.free), self.depth))
    bounds = [(B * i) // nparts for i in range(nparts + 1)]
    cut = lambda v, lo, hi: None if v is None else v[lo:hi]
    used, events = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
      k = self.free.pop(0)
      eng, stream = self.slots[k]
      stream.wait_stream(torch.cuda.current_stream())  # the feed tensors were produced there
      noise = feed.get('noise')  # box_model: [T, B, H, W]
      with torch.cuda.stream(stream):
        eng.forward(cut(feed['x'], lo, hi), d_in=cut(feed.get('d_in'), lo, hi), y_in=cut(feed.get('y_in'), lo, hi),
                    y_gt=cut(feed.get('y_gt'), lo, hi) if self.model.box_model else None,
                    noise=None if noise is None else noise[:, lo:hi], want_box='attn_box' in names)
        host = None
        if to_host:
          host = []
          for n in names:
            r = self.model._fetch(n, eng)
            host.append(torch.empty(r.shape, dtype=r.dtype, pin_memory=True).copy_(r, non_blocking=True))
        ev = torch.cuda.Event()
        ev.record(stream)
      used.append(k)
      events.append((ev, host))
    # `feed` (for a group: fresh concatenations made on the CURRENT stream; for a lone batch: the caller's tensors, which the
    # caller may drop as soon as submit() returns) is read by the engines' input copies on the slots' side streams, possibly
    # milliseconds from now (a slot queued behind another on its stream): the launch record keeps the tensors alive until the
    # launch has been waited for, so the caching allocator cannot hand their blocks to the next group's concatenation or to a
    # consumer's uploads while the copy is still queued
    rec = dict(used=used, events=events, bounds=bounds, refs=len(members), names=names, res=None, checked=False, feed=feed)
    lo = 0
    for m in members:
      m['launch'], m['lo'], m['hi'] = rec, lo, lo + m['B']
      lo += m['B']
      m['feed'] = None  # the engines hold their own copies

  def _finish(self, rec):
    """Wait for a launch; a starved controller is re-decoded (DecodeEngine.check_status) before anything is read."""
    if rec['checked']:
      return
    for k, (ev, host) in zip(rec['used'], rec['events']):
      eng, stream = self.slots[k]
      ev.synchronize()
      rec['feed'] = None  # the engines' input copies have run
      redone = eng.check_status()
      if host is not None and redone:  # the pinned copies were taken from the starved forward: copy the re-decoded outputs over them
        for h, n in zip(host, rec['names']):
          h.copy_(self.model._fetch(n, eng))
        torch.cuda.synchronize()
    rec['checked'] = True

  def _release(self, rec):
    rec['refs'] -= 1
    if rec['refs'] == 0:
      self.free.extend(rec['used'])
      rec['res'] = None

  def collect(self, as_numpy=False):
    """Results of the OLDEST batch in flight (blocks until it has finished), as model.run returns them."""
    if not self.pending:
      raise RecAttendError('collect() with no batch in flight')
    if self.pending[0]['launch'] is None:
      self._launch_group()  # it was still waiting for company
    t = self.pending.pop(0)
    rec, names = t['launch'], t['names']
    self._finish(rec)
    whole = t['lo'] == 0 and t['hi'] == rec['bounds'][-1]
    parts = []
    for (k, (ev, host)), plo, phi in zip(zip(rec['used'], rec['events']), rec['bounds'][:-1], rec['bounds'][1:]):
      lo, hi = max(t['lo'], plo) - plo, min(t['hi'], phi) - plo
      if hi <= lo:
        continue
      eng, stream = self.slots[k]
      sl = (lambda r: r) if (whole and len(rec['used']) == 1) else (lambda r: r[lo:hi])
      if host is not None:  # submit(to_host=True): already in pinned host memory
        res = [sl(h.numpy()) for h in host]
        as_numpy = True
      else:
        with torch.cuda.stream(stream):
          res = [sl(self.model._fetch(n, eng)) for n in names]
          if as_numpy:
            res = [r.detach().cpu().numpy() for r in res]
          elif rec['refs'] > 1 or not whole:
            res = [r.clone() for r in res]  # a slice of a buffer the next launch of this slot overwrites
        stream.synchronize()
      parts.append(res)
    self._release(rec)
    if len(parts) == 1:
      res = parts[0]
    elif as_numpy:
      res = [np.concatenate(col, axis=0) for col in zip(*parts)]
    else:
      res = [torch.cat(col, dim=0) for col in zip(*parts)]
    return res[0] if t['single'] else res

  def retire(self):
    """Wait for the OLDEST batch in flight and drop it without fetching (throughput measurement)."""
    if not self.pending:
      raise RecAttendError('retire() with no batch in flight')
    if self.pending[0]['launch'] is None:
      self._launch_group()
    t = self.pending.pop(0)
    for ev, _ in t['launch']['events']:
      ev.synchronize()
    t['launch']['feed'] = None
    self._release(t['launch'])

  def drain(self):
    """Wait for every batch in flight without fetching anything (throughput measurement)."""
    if self.group:
      # batches still waiting for company need a slot: the launched ones are older — let them finish and give theirs back first
      for _, stream in (self.slots or []):
        stream.synchronize()
      self.pending = [t for t in self.pending if t['launch'] is None]
      self.free = list(range(self.depth))
      self._launch_group()
    for _, stream in (self.slots or []):
      stream.synchronize()
    self.pending = []
    self.free = list(range(self.depth))


def _register_controller(model, opt, d, pt_ctrl=None):
  """Controller CNN, LSTM, glimpse MLP, controller MLP (full_model.py:263-409)."""
  T = d['T']
  relu = nn.relu
  ccnn = nn.cnn(d['ccnn_filters'], d['ccnn_channels'], d['ccnn_pool'],
                [relu] * d['ccnn_nlayers'], [d['use_bn']] * d['ccnn_nlayers'],
                phase_train=model.get('phase_train'), wd=opt['weight_decay'], scope='ctrl_cnn',
                model=model, init_weights=pt_ctrl and pt_ctrl.get('ccnn'))
  ccnn.declare_copies(T)
  Cf = d['ccnn_channels'][-1]
  cell = nn.lstm(Cf, d['hid'], wd=opt['weight_decay'], scope='ctrl_lstm', model=model,
                 init_weights=pt_ctrl and pt_ctrl.get('crnn'))
  gdims = [d['hid']] * d['n_gmlp'] + [d['G']]
  gmlp = nn.mlp(gdims, [relu] * (d['n_gmlp'] - 1) + [nn.softmax], add_bias=True,
                wd=opt['weight_decay'], scope='glimpse_mlp', model=model,
                init_weights=pt_ctrl and pt_ctrl.get('gmlp'))
  cdims = [d['hid']] + [d['mlp_dim']] * (d['n_cmlp'] - 1) + [9]
  cmlp = nn.mlp(cdims, [relu] * (d['n_cmlp'] - 1) + [None], add_bias=True,
                wd=opt['weight_decay'], scope='ctrl_mlp', model=model,
                init_weights=pt_ctrl and pt_ctrl.get('cmlp'))
  return ccnn, cell, gmlp, cmlp


def _load_pretrained(path):
  """The reference reads weights.h5 (full_model.py:271-284,...); h5py is not part of this
  stack, so the same key/value schema is read from an .npz archive."""
  if path is None:
    return None
  if str(path).endswith('.h5'):
    raise RecAttendError('HDF5 weight files are not readable here; convert to .npz with the '
                         'same keys (full_model_read.py:33-71)')
  return dict(np.load(path))


def _pretrained_groups(w, d, scopes):
  """Reshape a flat weight archive into nnlib's init_weights structures."""
  if w is None:
    return None
  T = d['T']

  def cnn_group(scope, nl):
    out = []
    for i in range(nl):
      g = {'w': w['%s_w_%d' % (scope, i)], 'b': w['%s_b_%d' % (scope, i)]}
      for t in range(T):
        for n in ('beta', 'gamma'):
          k = '%s_%d_%d_%s' % (scope, i, t, n)
          if k in w:
            g['%s_%d' % (n, t)] = w[k]
      out.append(g)
    return out

  res = {}
  if 'ctrl' in scopes:
    res['ccnn'] = cnn_group('ctrl_cnn', d['ccnn_nlayers'])
    res['crnn'] = {k: w['ctrl_lstm_' + k] for k in ('w_xi', 'w_hi', 'b_i', 'w_xf', 'w_hf', 'b_f',
                                                    'w_xu', 'w_hu', 'b_u', 'w_xo', 'w_ho', 'b_o')}
    res['gmlp'] = [{'w': w['glimpse_mlp_w_%d' % i], 'b': w['glimpse_mlp_b_%d' % i]}
                   for i in range(d['n_gmlp'])]
    res['cmlp'] = [{'w': w['ctrl_mlp_w_%d' % i], 'b': w['ctrl_mlp_b_%d' % i]}
                   for i in range(d['n_cmlp'])]
  if 'attn' in scopes:
    res['acnn'] = cnn_group('attn_cnn', d['acnn_nlayers'])
    res['adcnn'] = cnn_group('attn_dcnn', d['adcnn_nlayers'])
  if 'score' in scopes and 'score_mlp_w_0' in w:
    res['smlp'] = [{'w': w['score_mlp_w_0'], 'b': w['score_mlp_b_0']}]
  return res


def get_model(opt, is_training=True):
  """The attention model (full_model.py:13)."""
  d = derive_dims(opt)
  model = Model(opt, d)
  model['phase_train'] = {'value': False}
  pt_net = _get(opt, 'pretrain_net', None)
  w_ctrl = _load_pretrained(pt_net or _get(opt, 'pretrain_ctrl_net', None))
  w_attn = _load_pretrained(pt_net or _get(opt, 'pretrain_attn_net', None))
  w_all = _load_pretrained(pt_net)
  pt_ctrl = _pretrained_groups(w_ctrl, d, ('ctrl',))
  pt_attn = _pretrained_groups(w_attn, d, ('attn',))
  pt_score = _pretrained_groups(w_all, d, ('score',))
  relu = nn.relu
  wd = opt['weight_decay']

  ccnn, cell, gmlp, cmlp = _register_controller(model, opt, d, pt_ctrl)
  acnn = nn.cnn(d['acnn_filters'], d['acnn_channels'], d['acnn_pool'],
                [relu] * d['acnn_nlayers'], [d['use_bn']] * d['acnn_nlayers'],
                phase_train=model['phase_train'], wd=wd, scope='attn_cnn', model=model,
                init_weights=pt_attn and pt_attn['acnn'])
  acnn.declare_copies(d['T'])
  smlp = nn.mlp([d['hid'] + d['core_dim'], 1], [nn.sigmoid], wd=wd, scope='score_mlp',
                model=model, init_weights=pt_score and pt_score.get('smlp'))
  adcnn = nn.dcnn(d['adcnn_filters'], d['adcnn_channels'], d['adcnn_unpool'],
                  [relu] * d['adcnn_nlayers'], use_bn=[d['use_bn']] * d['adcnn_nlayers'],
                  skip_ch=d['skip_ch'], phase_train=model['phase_train'], wd=wd, model=model,
                  init_weights=pt_attn and pt_attn['adcnn'], scope='attn_dcnn')
  adcnn.declare_copies(d['T'])
  model.closures = dict(ccnn=ccnn, crnn_cell=cell, gmlp=gmlp, cmlp=cmlp, acnn=acnn, smlp=smlp,
                        adcnn=adcnn)
  model['global_step'] = 0.0
  model.engine = ra_engine.DecodeEngine(d, model)
  model.is_training = is_training
  return model
