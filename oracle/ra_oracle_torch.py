"""Second CPU oracle (TEST INFRASTRUCTURE ONLY): the same eval-mode forward and loss head as
oracle/ra_oracle.py, restated INDEPENDENTLY on torch primitives (float64, CPU) and differentiable.

Why it exists:
  * the neural path of the reference has no fixture to pin an oracle against ("parity unpinned",
    see ra_oracle.py); two restatements written against the reference's lines with different
    primitives (NumPy einsum / explicit loops there, torch.nn.functional here: F.conv2d,
    F.conv_transpose2d, F.max_pool2d, torch.matmul) that agree to 1e-9 are a stronger pin than
    either alone (tests/test_oracle_torch.py);
  * torch autograd through this restatement is the reference gradient the backward kernels of the
    training step (not built yet) will be checked against.

Only shape bookkeeping (`derive`) and the plain-C Hungarian are shared with ra_oracle.py; every
tensor operation below is written against the reference source again.  Nothing under
rec-attend-public_amd/ imports this module.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

import ra_oracle as ora

DT = torch.float64  # the checker's precision; bench.py's cpu_baseline leg times it at float32


def set_dtype(dt):
  """float64 (default: the parity / gradient checker) or float32 (the timed CPU stand-in, the
  reference's own arithmetic type)."""
  global DT
  DT = dt


def t64(a):
  if isinstance(a, torch.Tensor):
    return a.to(DT)
  return torch.as_tensor(np.asarray(a, dtype=np.float64)).to(DT)


# The product's mixed-precision training mode (model_opt['compute_dtype'] = 'bf16') is an extension the reference does
# not have.  Its oracle is THIS graph with the conv layers' three products taking bf16-rounded operands — the forward
# conv(R(x), R(w)), the data gradient convT(R(du), R(w)), the filter / bias gradients corr(R(x), R(du)), sum R(du) —
# everything else (and every sum) in the oracle's own precision.  R = round to nearest even, as v_cvt_pk_bf16_f32.
CONV_OPERANDS = None


def set_conv_operands(kind):
  """None (the reference's arithmetic), 'bf16' (bf16 operands of the three conv products) or 'bf16s' (the same, and the
  tensors between the conv layers' passes STORED as bf16, as the product's stacked training step keeps them: the
  pre-activation output U after its batch moments were taken, the activation Y of every layer but a net's last, and the
  gradients dY / dU — where the layer's channel count is one the product's float4 BatchNorm kernels take)."""
  global CONV_OPERANDS
  assert kind in (None, 'bf16', 'bf16s')
  CONV_OPERANDS = kind


def _rounds_operands():
  return CONV_OPERANDS in ('bf16', 'bf16s')


def _stores_bf16(cout):
  c4 = cout // 4
  return CONV_OPERANDS == 'bf16s' and cout % 4 == 0 and c4 & (c4 - 1) == 0 and c4 <= 64


def _bf16_round(t):
  return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _RoundOperand(torch.autograd.Function):
  """forward: R(t); backward: the gradient passes (the rounding belongs to the product it feeds, not to t)."""

  @staticmethod
  def forward(ctx, t):
    return _bf16_round(t)

  @staticmethod
  def backward(ctx, g):
    return g


class _RoundGradient(torch.autograd.Function):
  """forward: identity; backward: R(g) — the layer's output gradient as the backward products read it."""

  @staticmethod
  def forward(ctx, t):
    return t.view_as(t)

  @staticmethod
  def backward(ctx, g):
    return _bf16_round(g)


def conv_same(x, w, b):
  """nnlib.conv2d (nnlib.py:6-12): NHWC x, HWIO w, stride 1, SAME -> NHWC (odd kernels)."""
  k = w.shape[0]
  if _rounds_operands():
    x, w = _RoundOperand.apply(x), _RoundOperand.apply(w)
  y = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), padding=k // 2)
  y = y.permute(0, 2, 3, 1) + b
  return _RoundGradient.apply(y) if _rounds_operands() else y


def deconv_same(x, w, b, stride):
  """nnlib.py:372-376 conv2d_transpose, filter [f,f,out,in], SAME, output = input * stride.
  SURVEY.md §8a trap 2: stride 2, k = 3 is conv_transpose2d(padding=0)[..., :2n, :2n]; stride 1 is
  padding = 1."""
  if _rounds_operands():
    x, w = _RoundOperand.apply(x), _RoundOperand.apply(w)
  wt = w.permute(3, 2, 0, 1)  # [in, out, kh, kw]
  xi = x.permute(0, 3, 1, 2)
  if stride == 1:
    y = F.conv_transpose2d(xi, wt, stride=1, padding=w.shape[0] // 2)
  else:
    n_h, n_w = x.shape[1] * stride, x.shape[2] * stride
    y = F.conv_transpose2d(xi, wt, stride=stride, padding=0)[:, :, :n_h, :n_w]
  y = y.permute(0, 2, 3, 1) + b
  return _RoundGradient.apply(y) if _rounds_operands() else y


def pool(x, r):
  """nnlib.max_pool (nnlib.py:15-25); all sizes on the path are even."""
  if r == 1:
    return x
  return F.max_pool2d(x.permute(0, 3, 1, 2), r, r).permute(0, 2, 3, 1)


def bn_eval(x, P, key):
  """nnlib.batch_norm, eval branch (nnlib.py:113-119): EMA statistics, eps 1e-3 inside the sqrt."""
  inv = P[key + '_gamma'] * torch.rsqrt(P[key + '_ema_var'] + 1e-3)
  return x * inv + (P[key + '_beta'] - P[key + '_ema_mean'] * inv)


def bn_train(x, P, key, stats):
  """nnlib.batch_norm, training branch (nnlib.py:98-112): moments over (B, H, W) with the biased
  variance, differentiated through; the (mean, var) pair is recorded for the EMA update
  shadow = 0.9 shadow + 0.1 value (nnlib.py:103-110).  'bf16s': the moments are those of the conv's float32 output, the
  value normalised is its bf16 STORED form."""
  mean = x.mean(dim=(0, 1, 2))
  var = ((x - mean) ** 2).mean(dim=(0, 1, 2))
  if stats is not None:
    stats[key] = (mean.detach(), var.detach())
  if _stores_bf16(x.shape[-1]):
    x = _RoundOperand.apply(x)
  return (x - mean) * torch.rsqrt(var + 1e-3) * P[key + '_gamma'] + P[key + '_beta']


def _store_y(x, last):
  """'bf16s': a layer's activation is stored as bf16 (and so is its gradient on the way back), except a net's last
  output, which float32 kernels read (the controller, the score MLP / decoder input, the paste)."""
  if _BN['train'] and not last and _stores_bf16(x.shape[-1]):
    return _RoundGradient.apply(_RoundOperand.apply(x))
  return x


_BN = {'train': False, 'stats': None}


def bn(x, P, key):
  return bn_train(x, P, key, _BN['stats']) if _BN['train'] else bn_eval(x, P, key)


def cnn(x, P, scope, n, pools, tt, use_bn):
  """nnlib.cnn / run_cnn (nnlib.py:214-255): conv + b -> BN -> ReLU -> pool, all layers returned."""
  hs = []
  for i in range(n):
    h = conv_same(x, P['%s_w_%d' % (scope, i)], P['%s_b_%d' % (scope, i)])
    if use_bn:
      h = bn(h, P, '%s_%d_%d' % (scope, i, tt))
    x = _store_y(pool(torch.relu(h), pools[i]), i == n - 1)
    hs.append(x)
  return hs


def dcnn(x, P, scope, n, unpool, tt, skip, use_bn):
  """nnlib.dcnn / run_dcnn (nnlib.py:339-402): [concat skip] -> transposed conv + b -> BN -> ReLU."""
  hs = []
  for i in range(n):
    if skip is not None and skip[i] is not None:
      x = torch.cat([x, skip[i]], dim=3)
    h = deconv_same(x, P['%s_w_%d' % (scope, i)], P['%s_b_%d' % (scope, i)], unpool[i])
    if use_bn:
      h = bn(h, P, '%s_%d_%d' % (scope, i, tt))
    x = _store_y(torch.relu(h), i == n - 1)
    hs.append(x)
  return hs


def mlp(x, P, scope, acts):
  """nnlib.mlp / run_mlp (nnlib.py:476-493)."""
  hs = []
  for i, act in enumerate(acts):
    x = x @ P['%s_w_%d' % (scope, i)] + P['%s_b_%d' % (scope, i)]
    if act is not None:
      x = act(x)
    hs.append(x)
  return hs


def lstm(x, state, P, hid):
  """nnlib.lstm unroll (nnlib.py:637-649): state = [c | h]."""
  c, h = state[:, :hid], state[:, hid:]
  gate = lambda g: x @ P['ctrl_lstm_w_x' + g] + h @ P['ctrl_lstm_w_h' + g] + P['ctrl_lstm_b_' + g]
  gi, gf, go = torch.sigmoid(gate('i')), torch.sigmoid(gate('f')), torch.sigmoid(gate('o'))
  u = torch.tanh(gate('u'))
  c = gf * c + gi * u
  return torch.cat([c, go * torch.tanh(c)], dim=1)


def gaussian_filter(ctr, size, lg_var, L, Fn):
  """modellib.get_gaussian_filter (modellib.py:581-612) -> [B, L, F], not normalised."""
  j = torch.arange(Fn, dtype=DT)
  mu = ctr[:, None] + ((size[:, None] + 1.0) / Fn) * (j[None, :] - (Fn - 1) / 2.0)  # [B,F]
  l = torch.arange(L, dtype=DT)
  var = torch.exp(lg_var)[:, None, None]
  dd = l[None, :, None] - mu[:, None, :]
  return torch.exp(-0.5 * dd * dd / var) / (torch.sqrt(var) * math.sqrt(2 * math.pi))


def extract(x, fy, fx):
  """modellib.extract_patch (modellib.py:615-641): per channel F_y^T X F_x.  x [B,H,W,C]."""
  t = torch.einsum('blj,blwc->bjwc', fy, x)
  return torch.einsum('bjwc,bwi->bjic', t, fx)


def knob_setup(opt, y_gt, knobs, global_step):
  """full_model.py:559-625: the noisy ground-truth attention (get_gt_attn with the drawn padding /
  centre shift), the plain GT boxes for the greedy match, and the two knob masks."""
  T = y_gt.shape[1]
  mp = opt['padding'] + 4.0
  tl0, br0, box_gt = ora.get_gt_box(y_gt, padding_ratio=opt['attn_box_padding_ratio'], center_shift_ratio=0.0, min_padding=mp)
  tl, br, _ = ora.get_gt_box(y_gt, padding_ratio=knobs['pad'], center_shift_ratio=knobs['shift'], min_padding=mp)
  scale = 1.0 + np.log(1.0 + np.arange(T) * 3.0) if ora._opt(opt, 'knob_use_timescale', False) else np.ones(T)
  prob = lambda off: np.minimum(1.0, opt['knob_base'] * opt['knob_decay'] ** (
      max(0.0, global_step - off) / opt['steps_per_knob_decay']) * scale)[None, :, None]
  return {'ctr': t64((tl + br) / 2.0), 'size': t64(br - tl), 'box_gt': box_gt, 'tl_gt': tl0, 'br_gt': br0,
          'kb': t64((knobs['u_box'] <= prob(opt['knob_box_offset'])).astype(np.float64)),
          'ks': t64((knobs['u_segm'] <= prob(opt['knob_segm_offset'])).astype(np.float64)),
          'noise': t64(knobs['segm_noise']), 'y_gt': t64(y_gt)}


_KNOB = {'k': None}


def forward(opt, Pnp, x, d_in=None, y_in=None, requires_grad=(), phase_train=False, bn_stats=None, knobs=None,
            y_gt=None, global_step=0):
  """Forward of full_model.py:638-907 with use_knob False.  Returns (outputs dict of torch tensors,
  P dict); parameters named in `requires_grad` are leaves with gradients enabled.  phase_train:
  BatchNorm on batch statistics (recorded into `bn_stats` when a dict is passed)."""
  _BN['train'], _BN['stats'] = bool(phase_train), bn_stats
  _KNOB['k'] = None
  if knobs is not None and ora._opt(opt, 'use_knob', False):  # ground-truth mixing, training only
    kn = {k: np.asarray(v, dtype=np.float64) for k, v in knobs.items()}
    _KNOB['k'] = knob_setup(opt, np.asarray(y_gt, dtype=np.float64), kn, global_step)
  try:
    return _forward(opt, Pnp, x, d_in, y_in, requires_grad)
  finally:
    _BN['train'], _BN['stats'] = False, None
    _KNOB['k'] = None


def _forward(opt, Pnp, x, d_in, y_in, requires_grad):
  d = ora.derive(opt)
  P = {k: t64(v).requires_grad_(k in requires_grad) for k, v in Pnp.items()}
  x = t64(x)
  d_in = None if d_in is None else t64(d_in)
  y_in = None if y_in is None else t64(y_in)
  B, T, H, W, Fh, Fw, hid, G = x.shape[0], d['T'], d['H'], d['W'], d['Fh'], d['Fw'], d['hid'], d['G']
  canvas = torch.zeros((B, H, W, 1), dtype=DT)
  outs = {k: [] for k in ('y_out', 's_out', 'attn_box', 'attn_ctr', 'attn_size', 'x_patch', 'attn_ctr_norm',
                          'attn_lg_size')}

  def cat(flags):  # full_model.py:640-661: x, canvas, d_in, y_in in that order
    parts = [p for f, p in zip(flags, (x, canvas, d_in, y_in)) if f]
    return torch.cat(parts, dim=3)

  for tt in range(T):
    feat = cnn(cat(d['ctrl_in']), P, 'ctrl_cnn', d['ccnn_nlayers'], d['ccnn_pool'], tt, d['use_bn'])[-1]
    feat = feat.reshape(B, G, -1)
    state = torch.zeros((B, 2 * hid), dtype=DT)
    gmap = torch.full((B, G, 1), 1.0 / G, dtype=DT)
    for it in range(d['iters']):  # full_model.py:668-689
      state = lstm((feat * gmap).sum(dim=1), state, P, hid)
      h = state[:, hid:]
      if it < d['iters'] - 1:
        acts = [torch.relu] * (d['n_gmlp'] - 1) + [lambda z: torch.softmax(z, dim=1)]
        gmap = mlp(h, P, 'glimpse_mlp', acts)[-1][:, :, None]
    co = mlp(h, P, 'ctrl_mlp', [torch.relu] * (d['n_cmlp'] - 1) + [None])[-1]
    cn, ls = co[:, 0:2], co[:, 2:4]  # full_model.py:691-722
    if d['squash']:
      cn, ls = torch.tanh(cn), -F.softplus(ls)
    dims = torch.tensor([H, W], dtype=DT)
    ctr = (cn + 1.0) * dims / 2.0
    size = torch.exp(ls) * dims
    lg_var = torch.zeros_like(ctr) if d['fixed_var'] else torch.log(size) - torch.log(torch.tensor([Fh, Fw], dtype=DT))
    if d['dynamic_var']:
      lg_var = co[:, 4:6]
    if d['fixed_gamma']:
      attn_gamma, y_lg_gamma = torch.ones((B, 1, 1, 1), dtype=DT), torch.full((B, 1, 1, 1), 2.0, dtype=DT)
    else:
      attn_gamma, y_lg_gamma = torch.exp(co[:, 6]).reshape(B, 1, 1, 1), co[:, 8].reshape(B, 1, 1, 1)
    box_gamma = torch.exp(co[:, 7]).reshape(B, 1, 1, 1)
    fy, fx = gaussian_filter(ctr[:, 0], size[:, 0], lg_var[:, 0], H, Fh), \
        gaussian_filter(ctr[:, 1], size[:, 1], lg_var[:, 1], W, Fw)
    fyi, fxi = fy.transpose(1, 2), fx.transpose(1, 2)
    ones = torch.ones((B, Fh, Fw, 1), dtype=DT)
    attn_box = torch.sigmoid(extract(ones * box_gamma, fyi, fxi) - 5.0).reshape(B, 1, H, W)  # :738-741
    K = _KNOB['k']
    if K is not None:  # full_model.py:744-785: GT box kicked in, filters recomputed (lg_var unchanged)
      if ora._opt(opt, 'fixed_order', False):
        ctr_m, size_m, gm = K['ctr'][:, tt], K['size'][:, tt], None
      else:
        if ora._opt(opt, 'use_iou_box', False):  # modellib.f_iou_box on the corners (full_model.py:750-754)
          c_, s_ = ctr.detach().numpy(), size.detach().numpy()
          ta, ba = (c_ - s_ / 2.0)[:, None, :], (c_ + s_ / 2.0)[:, None, :]
          tb, bb = K['tl_gt'], K['br_gt']
          y1, x1 = np.maximum(ta[..., 0], tb[..., 0]), np.maximum(ta[..., 1], tb[..., 1])
          y2, x2 = np.minimum(ba[..., 0], bb[..., 0]), np.minimum(ba[..., 1], bb[..., 1])
          inter = (x1 < x2) * (y1 < y2) * (x2 - x1) * (y2 - y1)
          iou_t = (inter / ((ba[..., 1] - ta[..., 1]) * (ba[..., 0] - ta[..., 0]) +
                            (bb[..., 1] - tb[..., 1]) * (bb[..., 0] - tb[..., 0]) - inter))
          # the same IoU on the autograd tape: the reference stacks these per-timestep rows into the [B,T,T] matrix
          # its box matching AND its box loss use (full_model.py:931-934), differentiable through the corners
          tat, bat = (ctr - size / 2.0)[:, None, :], (ctr + size / 2.0)[:, None, :]
          tbt, bbt = t64(tb), t64(bb)
          y1t, x1t = torch.maximum(tat[..., 0], tbt[..., 0]), torch.maximum(tat[..., 1], tbt[..., 1])
          y2t, x2t = torch.minimum(bat[..., 0], bbt[..., 0]), torch.minimum(bat[..., 1], bbt[..., 1])
          it = (x1t < x2t).to(DT) * (y1t < y2t).to(DT) * (x2t - x1t) * (y2t - y1t)
          outs.setdefault('iou_box_steps', []).append(
              (it / ((bat[..., 1] - tat[..., 1]) * (bat[..., 0] - tat[..., 0]) +
                     (bbt[..., 1] - tbt[..., 1]) * (bbt[..., 0] - tbt[..., 0]) - it))[:, None, :])
        else:
          a = attn_box.detach().numpy()
          iou_t = ora.f_inter(a, K['box_gt']) / ora.f_union(a, K['box_gt'], eps=1e-5)
        gm = t64(ora.f_greedy_match(iou_t, np.zeros_like(iou_t)))
        ctr_m, size_m = (gm[:, :, None] * K['ctr']).sum(dim=1), (gm[:, :, None] * K['size']).sum(dim=1)
      kb = K['kb'][:, tt]
      ctr = kb * ctr_m + (1 - kb) * ctr
      size = kb * size_m + (1 - kb) * size
      fy, fx = gaussian_filter(ctr[:, 0], size[:, 0], lg_var[:, 0], H, Fh), \
          gaussian_filter(ctr[:, 1], size[:, 1], lg_var[:, 1], W, Fw)
      fyi, fxi = fy.transpose(1, 2), fx.transpose(1, 2)
    x_patch = attn_gamma * extract(cat(d['attn_in']), fy, fx)                              # :788-789
    h_acnn = cnn(x_patch, P, 'attn_cnn', d['acnn_nlayers'], d['acnn_pool'], tt, d['use_bn'])
    h_core = h_acnn[-1].reshape(B, -1)                                                        # :794
    skip = None
    if d['add_skip_conn']:                                                                    # :798-805
      rev = h_acnn[::-1][1:] + [x_patch]
      skip = [None] + [hh if sk else None for sk, hh in zip(d['skip_rev'], rev)]
    y_patch = dcnn(h_acnn[-1], P, 'attn_dcnn', d['adcnn_nlayers'], d['adcnn_unpool'], tt, skip, d['use_bn'])[-1]
    y = torch.sigmoid(torch.exp(y_lg_gamma) * extract(y_patch, fyi, fxi) - 5.0).reshape(B, 1, H, W)  # :810-815
    if d['disable_overwrite']:
      y = (1 - canvas).reshape(B, 1, H, W) * y
    s = torch.sigmoid(torch.cat([h, h_core], dim=1) @ P['score_mlp_w_0'] + P['score_mlp_b_0'])  # :821-822
    y_c = y.reshape(B, H, W, 1)
    if K is not None:  # :826-841: GT segmentation (with noise) kicked in for the canvas
      gsel = K['y_gt'][:, tt] if gm is None else (gm[:, :, None, None] * K['y_gt']).sum(dim=1)
      gsel = gsel - gsel * K['noise'][tt]
      ks = K['ks'][:, tt].reshape(B, 1, 1, 1)
      y_c = ks * gsel[..., None] + (1 - ks) * y_c
    canvas = torch.maximum(y_c, canvas)                                                        # :843-848
    if ora._opt(opt, 'stop_canvas_grad', True):
      canvas = canvas.detach()
    for k, v in (('y_out', y), ('s_out', s), ('attn_box', attn_box), ('attn_ctr', ctr[:, None]),
                 ('attn_size', size[:, None]), ('x_patch', x_patch[:, None]), ('attn_ctr_norm', cn[:, None]),
                 ('attn_lg_size', ls[:, None])):
      outs[k].append(v)
  res = {k: torch.cat(v, dim=1) for k, v in outs.items()}
  res['canvas'] = canvas
  return res, P


def iou_pairwise(a, b):
  """modellib.f_iou pairwise (modellib.py:124-155), eps summed per pixel (:119-122)."""
  inter = torch.einsum('bnhw,bmhw->bnm', a, b)
  sa, sb = a.sum(dim=(2, 3))[:, :, None], b.sum(dim=(2, 3))[:, None, :]
  return inter / (sa + sb - inter + 1e-5 * a.shape[2] * a.shape[3])


def weight_decay_term(opt, P):
  """The wd * l2_loss(w) terms nnlib.weight_variable adds to the 'losses' collection
  (nnlib.py:59-61): l2_loss = sum(w^2) / 2, filters and matrices only."""
  wd = ora._opt(opt, 'weight_decay', 0.0) or 0.0
  tot = 0.0
  for k, v in P.items():
    tail = k.split('_')
    if 'w' in tail or (len(tail[-1]) == 3 and tail[-1][0] == 'w' and tail[-1][1] in 'xh'):
      tot = tot + wd * 0.5 * (v ** 2).sum()
  return tot


def loss_head(opt, fwd, y_gt, s_gt):
  """full_model.py:913-1025 for box_loss_fn = segm_loss_fn = 'iou' (the run scripts' setting): the
  differentiable total loss plus the pieces.  The matchings are constants (Hungarian,
  ops.NoGradient, modellib.py:11)."""
  y_gt, s_gt = t64(y_gt), t64(s_gt)
  B, T = s_gt.shape
  _, _, box_gt = ora.get_gt_box(y_gt.numpy(), padding_ratio=opt['attn_box_padding_ratio'],
                                center_shift_ratio=0.0, min_padding=opt['padding'] + 4.0)
  box_gt = t64(box_gt)

  def match_of(iou):
    if ora._opt(opt, 'fixed_order', False):
      return t64(ora.get_identity_match(s_gt.numpy()))
    return t64(ora.f_segm_match(iou.detach().numpy(), s_gt.numpy()))

  out = {}
  iou_box = fwd['iou_box_steps'] if 'iou_box_steps' in fwd else iou_pairwise(fwd['attn_box'], box_gt)  # :928-936
  m_box = match_of(iou_box)
  cnt_box = torch.clamp(m_box.sum(dim=(1, 2)), min=1.0)
  fixed = bool(ora._opt(opt, 'fixed_order', False))
  # fixed order: f_iou(pairwise=False) summed over ALL T, unmasked, over the identity match's count (:922-924,941-945)
  masked = lambda iou, m: torch.diagonal(iou, dim1=1, dim2=2).sum(dim=1) if fixed else (iou * m).sum(dim=(1, 2))
  out['iou_soft_box'] = (masked(iou_box, m_box) / cnt_box).sum() / B
  iou = iou_pairwise(fwd['y_out'], y_gt)
  m = match_of(iou)
  cnt = torch.clamp(m.sum(dim=(1, 2)), min=1.0)
  out['iou_soft'] = (masked(iou, m) / cnt).sum() / B
  s_out = fwd['s_out']
  s_min = torch.cummin(s_out, dim=1)[0]                                   # modellib.py:40-53
  s_max = torch.flip(torch.cummax(torch.flip(s_out, [1]), dim=1)[0], [1])  # :56-68
  ms = m.sum(dim=2)
  bce = -ms * torch.log(s_min + 1e-5) - (1 - ms) * torch.log(1 - s_max + 1e-5)  # :430-437
  out['conf_loss'] = bce.sum() / B / T
  box_loss, segm_loss = -out['iou_soft_box'], -out['iou_soft']
  blf = ora._opt(opt, 'box_loss_fn', 'iou')
  if blf in ('mse', 'huber'):  # f_match_loss on (ctr_norm, lg_size) (full_model.py:891-892,952-964; modellib.py:440-478)
    tl, br, _ = ora.get_gt_box(y_gt.numpy(), padding_ratio=opt['attn_box_padding_ratio'], center_shift_ratio=0.0,
                               min_padding=opt['padding'] + 4.0)
    hw = np.array([opt['inp_height'], opt['inp_width']], dtype=np.float64)
    pgt = t64(np.concatenate([((tl + br) / 2.0) / (hw / 2.0) - 1.0, np.log((br - tl) / hw)], axis=2))
    p = torch.cat([fwd['attn_ctr_norm'], fwd['attn_lg_size']], dim=2)
    err = p[:, :, None, :] - pgt[:, None, :, :]
    if blf == 'mse':
      pair = 0.5 * err * err
    else:  # modellib.py:514-522: indicator err <= 1
      ind = (err <= 1).to(err.dtype)
      pair = 0.5 * err * err * ind + (err.abs() - 0.5) * (1 - ind)
    box_loss = ((pair.sum(dim=3) * m_box).sum(dim=(1, 2)) / cnt_box).sum() / B / 4.0
  if ora._opt(opt, 'segm_loss_fn', 'iou') == 'wt_cov':  # modellib.py:277-302
    sg = y_gt.sum(dim=(2, 3))
    wts = sg / (sg.sum(dim=1, keepdim=True) + (sg == 0).to(sg.dtype))
    segm_loss = -(iou.max(dim=1)[0] * wts).sum() / B
  out['box_loss'], out['segm_loss'] = box_loss, segm_loss
  out['loss'] = box_loss + segm_loss + ora._opt(opt, 'loss_mix_ratio', 1.0) * out['conf_loss']
  out['match'], out['match_box'] = m, m_box
  return out


def box_forward_loss(opt, Pnp, x, y_gt, s_gt, noise, requires_grad=(), phase_train=True, bn_stats=None, d_in=None, y_in=None):
  """box_model.py:403-652 restated differentiably: the controller-only model, its canvas always
  teacher-forced from the greedily matched ground truth times (1 - noise[tt]) (noise [T,B,H,W], the
  draws of :500-502), box loss (matched soft IoU; 'mse' / 'huber' on (centre, log size)) + conf loss
  (+ the caller adds weight decay).  d_in / y_in: the KITTI / Cityscapes stage-1 inputs, concatenated behind
  (x, canvas) as box_model.py:404-410 does."""
  _BN['train'], _BN['stats'] = bool(phase_train), bn_stats
  try:
    d = ora.derive(opt, box_model=True)
    P = {k: t64(v).requires_grad_(k in requires_grad) for k, v in Pnp.items()}
    x, y_gt_t, s_gt_t, noise = t64(x), t64(y_gt), t64(s_gt), t64(noise)
    B, T, H, W, Fh, Fw, hid, G = x.shape[0], d['T'], d['H'], d['W'], d['Fh'], d['Fw'], d['hid'], d['G']
    tl, br, box_gt_np = ora.get_gt_box(np.asarray(y_gt, dtype=np.float64), padding_ratio=opt['attn_box_padding_ratio'],
                                       center_shift_ratio=0.0)
    box_gt = t64(box_gt_np)
    canvas = torch.zeros((B, H, W, 1), dtype=DT)
    boxes, ss, cns, lss = [], [], [], []
    dims = torch.tensor([H, W], dtype=DT)
    more = [t64(d_in), t64(y_in)] if d['add_d_out'] else []  # box_model.py:406-409
    for tt in range(T):
      feat = cnn(torch.cat([x, canvas] + more, dim=3), P, 'ctrl_cnn', d['ccnn_nlayers'], d['ccnn_pool'], tt, d['use_bn'])[-1]
      feat = feat.reshape(B, G, -1)
      state = torch.zeros((B, 2 * hid), dtype=DT)
      gmap = torch.full((B, G, 1), 1.0 / G, dtype=DT)
      for it in range(d['iters']):
        state = lstm((feat * gmap).sum(dim=1), state, P, hid)
        h = state[:, hid:]
        if it < d['iters'] - 1:
          acts = [torch.relu] * (d['n_gmlp'] - 1) + [lambda z: torch.softmax(z, dim=1)]
          gmap = mlp(h, P, 'glimpse_mlp', acts)[-1][:, :, None]
      co = mlp(h, P, 'ctrl_mlp', [torch.relu] * (d['n_cmlp'] - 1) + [None])[-1]
      cn, ls = co[:, 0:2], co[:, 2:4]
      if d['squash']:
        cn, ls = torch.tanh(cn), -F.softplus(ls)
      ctr, size = (cn + 1.0) * dims / 2.0, torch.exp(ls) * dims
      lg_var = torch.zeros_like(ctr) if d['fixed_var'] else torch.log(size) - torch.log(torch.tensor([Fh, Fw], dtype=DT))
      if d['dynamic_var']:
        lg_var = co[:, 4:6]
      fy, fx = gaussian_filter(ctr[:, 0], size[:, 0], lg_var[:, 0], H, Fh), \
          gaussian_filter(ctr[:, 1], size[:, 1], lg_var[:, 1], W, Fw)
      box = torch.sigmoid(torch.exp(co[:, 7]).reshape(B, 1, 1) * fy.sum(dim=2)[:, :, None] * fx.sum(dim=2)[:, None, :] - 5.0)
      if ora._opt(opt, 'fixed_order', False):
        ysel = y_gt_t[:, tt]
      else:
        a = box.detach().numpy()[:, None]
        gm = t64(ora.f_greedy_match(ora.f_inter(a, box_gt_np) / ora.f_union(a, box_gt_np, eps=1e-5), np.zeros((B, T))))
        ysel = (gm[:, :, None, None] * y_gt_t).sum(dim=1)
      ysel = ysel - ysel * noise[tt]
      canvas = torch.maximum(ysel[..., None], canvas).detach()
      s = h @ P['score_mlp_w_0'] + P['score_mlp_b_0']
      s = torch.sigmoid(s) if d['nsc'] == 1 else torch.softmax(s, dim=1)
      boxes.append(box)
      ss.append(s[:, None])
      cns.append(cn[:, None])
      lss.append(ls[:, None])
    attn_box, s_out = torch.stack(boxes, dim=1), torch.cat(ss, dim=1)
    iou = iou_pairwise(attn_box, box_gt)
    if ora._opt(opt, 'fixed_order', False):
      m = t64(ora.get_identity_match(np.asarray(s_gt, np.float64)))
    else:
      m = t64(ora.f_segm_match(iou.detach().numpy(), np.asarray(s_gt, np.float64)))
    cnt = torch.clamp(m.sum(dim=(1, 2)), min=1.0)
    iou_soft_box = ((iou * m).sum(dim=(1, 2)) / cnt).sum() / B
    box_loss = -iou_soft_box
    blf = ora._opt(opt, 'box_loss_fn', 'iou')
    if blf in ('mse', 'huber'):
      hw = np.array([H, W], dtype=np.float64)
      pgt = t64(np.concatenate([((tl + br) / 2.0) / (hw / 2.0) - 1.0, np.log((br - tl) / hw)], axis=2))
      err = torch.cat([torch.cat(cns, dim=1), torch.cat(lss, dim=1)], dim=2)[:, :, None, :] - pgt[:, None, :, :]
      if blf == 'mse':
        pair = 0.5 * err * err
      else:
        ind = (err <= 1).to(err.dtype)
        pair = 0.5 * err * err * ind + (err.abs() - 0.5) * (1 - ind)
      box_loss = ((pair.sum(dim=3) * m).sum(dim=(1, 2)) / cnt).sum() / B / 4.0
    sc = s_out[:, :, 0] if d['nsc'] == 1 else 1 - s_out[:, :, 0]                         # box_model.py:620-625
    s_min = torch.cummin(sc, dim=1)[0]
    s_max = torch.flip(torch.cummax(torch.flip(sc, [1]), dim=1)[0], [1])
    ms = m.sum(dim=2)
    conf = (-ms * torch.log(s_min + 1e-5) - (1 - ms) * torch.log(1 - s_max + 1e-5)).sum() / B / T
    return {'loss': box_loss + conf, 'box_loss': box_loss, 'conf_loss': conf, 'iou_soft_box': iou_soft_box,
            'match_box': m, 's_out': s_out, 'attn_box': attn_box}, P
  finally:
    _BN['train'], _BN['stats'] = False, None
