"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the recurrent-attention decode loop.

A NumPy restatement of the forward (eval-mode) path of the reference's
``full_model.get_model`` / ``box_model.get_model`` and of the ``nnlib`` / ``modellib``
operators that path uses.  Every function cites the reference ``file:line`` it follows
(paths relative to the reference checkout).

PARITY STATUS: **parity unpinned** for the neural path (cross-checked by a second, independent
torch restatement, oracle/ra_oracle_torch.py, to 1e-9).  The reference ships no test,
fixture or golden vector for nnlib/modellib/full_model (its only test file is
``hungarian_tf_tests.py``) and its runtime (Python 2.7 + TensorFlow 0.12) is not
installable here, so this restatement cannot be checked against reference outputs.
It is pinned instead by (i) line-by-line restatement, (ii) an independent
``torch.nn.functional`` cross-check (tests/test_oracle.py) and (iii) analytic
known-answer tests.  The third-party arithmetic the reference delegates to is
TensorFlow 0.12 (README.md:6; Conv2D, MaxPool, Conv2DBackpropInput, BatchMatMul,
MatMul, tf.nn.batch_normalization) whose published semantics are restated below.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``rec-attend-public_amd/``) never does.

All tensors are NHWC like the reference.  ``dtype`` selects float64 (default, the
checker) or float32.
"""
import numpy as np

# --------------------------------------------------------------------------------------
# nnlib operators
# --------------------------------------------------------------------------------------


def _same_pad(size, k, stride):
  """TensorFlow 'SAME' padding: out = ceil(size/stride); extra pad goes at the end."""
  out = -(-size // stride)
  total = max((out - 1) * stride + k - size, 0)
  lo = total // 2
  return out, lo, total - lo


def conv2d(x, w, stride=1):
  """nnlib.py:6-12 — tf.nn.conv2d(x, w, [1,s,s,1], 'SAME'); x [B,H,W,Ci], w [F,F,Ci,Co].

  Cross-correlation (no kernel flip), zero padding.
  """
  B, H, W, Ci = x.shape
  F = w.shape[0]
  Ho, pt, pb = _same_pad(H, F, stride)
  Wo, pl, pr = _same_pad(W, F, stride)
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  out = np.zeros((B, Ho, Wo, w.shape[3]), dtype=x.dtype)
  for ky in range(F):
    for kx in range(F):
      sl = xp[:, ky:ky + (Ho - 1) * stride + 1:stride, kx:kx + (Wo - 1) * stride +
              1:stride, :]
      out += sl @ w[ky, kx]
  return out


def max_pool(x, ratio):
  """nnlib.py:15-25 — tf.nn.max_pool ksize=stride=ratio, 'SAME' (-inf padding)."""
  B, H, W, C = x.shape
  Ho, pt, pb = _same_pad(H, ratio, ratio)
  Wo, pl, pr = _same_pad(W, ratio, ratio)
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), constant_values=-np.inf)
  xp = xp.reshape(B, Ho, ratio, Wo, ratio, C)
  return xp.max(axis=(2, 4))


def batch_norm_eval(x, beta, gamma, ema_mean, ema_var, eps=1e-3):
  """nnlib.py:113-119 with phase_train=False.

  tf.nn.batch_normalization(x, mean, var, beta, gamma, 1e-3) as published for TF 0.12:
  inv = rsqrt(var + eps) * gamma ; y = x * inv + (beta - mean * inv).
  """
  inv = gamma / np.sqrt(ema_var + eps)
  return x * inv + (beta - ema_mean * inv)


def relu(x):
  return np.maximum(x, 0)


def sigmoid(x):
  return 1.0 / (1.0 + np.exp(-x))


def softmax(x):
  """tf.nn.softmax over the last axis (max-subtracted, as TF's kernel does)."""
  e = np.exp(x - x.max(axis=-1, keepdims=True))
  return e / e.sum(axis=-1, keepdims=True)


def conv2d_transpose(x, w, stride):
  """nnlib.py:372-376 — tf.nn.conv2d_transpose(x, w[F,F,Co,Ci], out=[B,H*s,W*s,Co], 'SAME').

  Defined by TF as the gradient of conv2d('SAME', stride) w.r.t. its input: the forward
  conv over an image of size n*s has out[o] = sum_k in[o*s + k - pad_lo]*w[k], so the
  transpose scatters x[i] * w[k] to position i*s + k - pad_lo.
  """
  B, H, W, Ci = x.shape
  F, _, Co, Ci2 = w.shape
  assert Ci == Ci2
  Ho, Wo = H * stride, W * stride
  _, pt, _ = _same_pad(Ho, F, stride)
  _, pl, _ = _same_pad(Wo, F, stride)
  full = np.zeros((B, (H - 1) * stride + F, (W - 1) * stride + F, Co), dtype=x.dtype)
  for ky in range(F):
    for kx in range(F):
      full[:, ky:ky + (H - 1) * stride + 1:stride, kx:kx + (W - 1) * stride +
           1:stride, :] += x @ w[ky, kx].T
  # crop: output index o corresponds to full index o + pad_lo
  out = np.zeros((B, Ho, Wo, Co), dtype=x.dtype)
  fy = full[:, pt:pt + Ho, pl:pl + Wo, :]
  out[:, :fy.shape[1], :fy.shape[2], :] = fy
  return out


def run_cnn(x, P, scope, nlayers, pool, copy, use_bn=True):
  """nnlib.py:214-255 — conv+b -> BN(copy) -> relu -> maxpool; returns list of layers."""
  h = []
  prev = x
  for ii in range(nlayers):
    a = conv2d(prev, P['%s_w_%d' % (scope, ii)]) + P['%s_b_%d' % (scope, ii)]
    if use_bn:
      k = '%s_%d_%d_' % (scope, ii, copy)
      a = batch_norm_eval(a, P[k + 'beta'], P[k + 'gamma'], P[k + 'ema_mean'],
                          P[k + 'ema_var'])
    a = relu(a)
    if pool[ii] > 1:
      a = max_pool(a, pool[ii])
    h.append(a)
    prev = a
  return h


def run_dcnn(x, P, scope, nlayers, unpool, copy, skip=None, use_bn=True):
  """nnlib.py:339-402 — [concat skip] -> conv2d_transpose+b -> BN(copy) -> relu."""
  h = []
  prev = x
  for ii in range(nlayers):
    if skip is not None and skip[ii] is not None:
      prev = np.concatenate([prev, skip[ii]], axis=3)
    a = conv2d_transpose(prev, P['%s_w_%d' % (scope, ii)], unpool[ii]) + \
        P['%s_b_%d' % (scope, ii)]
    if use_bn:
      k = '%s_%d_%d_' % (scope, ii, copy)
      a = batch_norm_eval(a, P[k + 'beta'], P[k + 'gamma'], P[k + 'ema_mean'],
                          P[k + 'ema_var'])
    a = relu(a)
    h.append(a)
    prev = a
  return h


def run_mlp(x, P, scope, acts):
  """nnlib.py:476-493 — act(x W + b) per layer; returns list."""
  h = []
  prev = x
  for ii, act in enumerate(acts):
    a = prev @ P['%s_w_%d' % (scope, ii)] + P['%s_b_%d' % (scope, ii)]
    if act is not None:
      a = act(a)
    h.append(a)
    prev = a
  return h


def lstm_step(inp, state, P, scope, hid):
  """nnlib.py:637-649 — state = [c | h] (c first); returns (state, g_i, g_f, g_o)."""
  c = state[:, :hid]
  h = state[:, hid:]
  g = lambda n: inp @ P['%s_w_x%s' % (scope, n)] + h @ P['%s_w_h%s' % (scope, n)] + \
      P['%s_b_%s' % (scope, n)]
  g_i = sigmoid(g('i'))
  g_f = sigmoid(g('f'))
  g_o = sigmoid(g('o'))
  u = np.tanh(g('u'))
  c = g_f * c + g_i * u
  h = g_o * np.tanh(c)
  return np.concatenate([c, h], axis=1), g_i, g_f, g_o


# --------------------------------------------------------------------------------------
# modellib operators
# --------------------------------------------------------------------------------------


def get_gaussian_filter(center, size, lg_var, image_size, filter_size):
  """modellib.py:581-612 — un-normalised DRAW filter bank, returns [B, L, F]."""
  dt = center.dtype
  span_filter = np.arange(filter_size, dtype=dt).reshape(1, 1, -1)
  center = center.reshape(-1, 1, 1)
  size = size.reshape(-1, 1, 1)
  mu = center + (size + 1) / filter_size * (span_filter - (filter_size - 1) / 2.0)
  lg_var = lg_var.reshape(-1, 1, 1)
  span = np.arange(image_size, dtype=dt).reshape(1, image_size, 1)
  return (1 / np.sqrt(np.exp(lg_var)) / np.sqrt(dt.type(2 * np.pi))) * \
      np.exp(-0.5 * (span - mu) * (span - mu) / np.exp(lg_var))


def extract_patch(x, f_y, f_x, nchannels):
  """modellib.py:615-641 — per channel f_y^T . x_d . f_x ; x [B,H,W,D] -> [B,FH,FW,D]."""
  out = []
  for d in range(nchannels):
    xc = x[:, :, :, d]
    out.append(np.matmul(np.matmul(np.transpose(f_y, (0, 2, 1)), xc), f_x)[..., None])
  return np.concatenate(out, axis=3)


def get_unnormalized_attn(ctr_norm, lg_size, H, W):
  """modellib.py:843-847 (:752-764, :812-825)."""
  img = np.array([H, W], dtype=ctr_norm.dtype)
  return (ctr_norm + 1.0) * (img / 2.0), np.exp(lg_size) * img


def get_normalized_var(size, fh, fw):
  """modellib.py:782-793."""
  return np.log(size) - np.log(np.array([fh, fw], dtype=size.dtype))


def f_inter(a, b):
  """modellib.py:104-107."""
  return (a * b).sum(axis=(-2, -1))


def f_union(a, b, eps=1e-5):
  """modellib.py:110-114."""
  return (a + b - a * b + eps).sum(axis=(-2, -1))


def f_greedy_match(score, matched):
  """modellib.py:366-379."""
  score = score * (1.0 - matched)
  mx = score.max(axis=1, keepdims=True)
  match = (score == mx).astype(score.dtype)
  return match / match.sum(axis=1, keepdims=True)


def get_gt_box(y_gt, padding_ratio=0.0, center_shift_ratio=0.0, min_padding=10.0):
  """modellib.py:663-701 — y_gt [B,T,H,W] -> top_left, bot_right [B,T,2], box [B,T,H,W]."""
  dt = y_gt.dtype
  B, T, H, W = y_gt.shape
  idx_y = np.broadcast_to(np.arange(H, dtype=dt).reshape(1, 1, H, 1), (B, T, H, W))
  idx_x = np.broadcast_to(np.arange(W, dtype=dt).reshape(1, 1, 1, W), (B, T, H, W))
  idx = np.stack([idx_y, idx_x], axis=4)
  nz = (y_gt.sum(axis=(2, 3)) > 0).astype(dt)[:, :, None]
  idx_min = idx + ((1.0 - y_gt) * dt.type(H * W))[..., None]
  idx_max = idx * y_gt[..., None]
  top_left = idx_min.min(axis=(2, 3))
  bot_right = idx_max.max(axis=(2, 3))
  size = bot_right - top_left
  top_left = top_left + center_shift_ratio * size
  top_left = top_left - np.maximum(padding_ratio * size, min_padding)
  bot_right = bot_right + center_shift_ratio * size
  bot_right = bot_right + np.maximum(padding_ratio * size, min_padding)
  tl = top_left.reshape(B, T, 1, 1, 2)
  br = bot_right.reshape(B, T, 1, 1, 2)
  box = (idx >= tl).astype(dt).prod(axis=4) * (idx <= br).astype(dt).prod(axis=4)
  top_left = top_left * nz
  bot_right = nz * bot_right + (1 - nz) * (2 * min_padding)
  return top_left, bot_right, box


def f_segm_match_precondition(iou, s_gt):
  """modellib.py:395-405 — masked, quantised IoU handed to the Hungarian op (+1e-5).

  tf.round in TF 0.12 is taken as floor(x + 0.5) (SURVEY.md §8a trap 9; unverifiable).
  """
  mask_x = s_gt[:, None, :]
  mask_y = s_gt[:, :, None]
  iou_mask = iou * mask_x * mask_y
  iou_mask = np.floor(iou_mask * 1e6 + 0.5) / 1e6
  return (iou_mask + 1e-5).astype(np.float32), mask_x, mask_y


# --------------------------------------------------------------------------------------
# loss / statistics head of the training graph (forward only), full_model.py:913-1097
# --------------------------------------------------------------------------------------

_HUNG = None


def hungarian_c(w):
  """The plain-C Hungarian oracle (oracle/hungarian_oracle.c, restating hungarian.cc) on a
  float32 [B,N,M] weight tensor -> matching [B,N,M]."""
  global _HUNG
  import ctypes
  import os
  if _HUNG is None:
    _HUNG = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                     'libhungarian_oracle.so'))
    _HUNG.ora_hungarian_f32.restype = ctypes.c_int
  w = np.ascontiguousarray(w, np.float32)
  B, N, M = w.shape
  m, cx, cy = np.zeros_like(w), np.zeros((B, N), np.float32), np.zeros((B, M), np.float32)
  p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
  rc = _HUNG.ora_hungarian_f32(p(w), B, N, M, p(m), p(cx), p(cy))
  if rc < 0:
    raise RuntimeError('hungarian oracle failed: %d' % rc)
  return m


def f_iou_pairwise(a, b):
  """modellib.py:124-155 pairwise=True — a [B,N,H,W], b [B,M,H,W] -> [B,N,M];
  f_inter :113-116, f_union :119-122 (the 1e-5 is summed over every pixel)."""
  inter = np.einsum('bnhw,bmhw->bnm', a, b)
  sa = a.sum(axis=(2, 3))[:, :, None]
  sb = b.sum(axis=(2, 3))[:, None, :]
  hw = a.shape[2] * a.shape[3]
  return inter / (sa + sb - inter + 1e-5 * hw)


def f_dice_pairwise(a, b):
  """modellib.py:71-104 pairwise=True — 2 * inter / (sum(a + 1e-5) + sum(b + 1e-5))."""
  inter = np.einsum('bnhw,bmhw->bnm', a, b)
  hw = a.shape[2] * a.shape[3]
  ca = a.sum(axis=(2, 3))[:, :, None] + 1e-5 * hw
  cb = b.sum(axis=(2, 3))[:, None, :] + 1e-5 * hw
  return 2 * inter / (ca + cb)


def get_identity_match(s_gt):
  """modellib.py:28-37."""
  T = s_gt.shape[1]
  return np.eye(T, dtype=s_gt.dtype)[None] * s_gt[:, None, :] * s_gt[:, :, None]


def f_segm_match(iou, s_gt):
  """modellib.py:382-415 (Hungarian on the masked, quantised IoU, masked again)."""
  w, mask_x, mask_y = f_segm_match_precondition(iou, s_gt)
  return hungarian_c(w).astype(iou.dtype) * mask_x * mask_y


def f_coverage_weight(y_gt):
  """modellib.py:277-289."""
  s = y_gt.sum(axis=(2, 3))
  tot = s.sum(axis=1, keepdims=True) + (s == 0).astype(s.dtype)
  return s / tot


def f_weighted_coverage(iou, y_gt):
  """modellib.py:292-302 (coverage = max over the output axis, :265-274)."""
  return (iou.max(axis=1) * f_coverage_weight(y_gt)).sum() / y_gt.shape[0]


def f_unweighted_coverage(iou, count):
  """modellib.py:305-313."""
  return (iou.max(axis=1).sum(axis=1) / count).sum() / iou.shape[0]


def f_conf_loss(s_out, match):
  """modellib.py:316-339 with use_cum_min=True; f_cum_min :40-53, f_cum_max :56-68,
  f_bce_minmax :430-437."""
  B, T = s_out.shape
  match_sum = match.sum(axis=2)
  s_min = np.minimum.accumulate(s_out, axis=1)
  s_max = np.maximum.accumulate(s_out[:, ::-1], axis=1)[:, ::-1]
  eps = 1e-5
  bce = -match_sum * np.log(s_min + eps) - (1 - match_sum) * np.log(1 - s_max + eps)
  return bce.sum() / B / T


def f_count_stats(s_out, s_gt):
  """f_count_acc modellib.py:482-494, f_dic :497-511 (abs False / True)."""
  B = s_out.shape[0]
  cout = (s_out > 0.5).astype(s_out.dtype).sum(axis=1)
  cgt = s_gt.sum(axis=1)
  return (cout == cgt).astype(s_out.dtype).sum() / B, (cout - cgt).sum() / B, \
      np.abs(cout - cgt).sum() / B


def loss_head(opt, fwd, y_gt, s_gt):
  """The loss and statistics part of the training graph, full_model.py:913-1097, evaluated on
  the outputs `fwd` of full_model_forward (phase_train False: the GT knobs are off, :762-769,
  :839-841, and the per-timestep iou_soft_box of the use_knob branch, :756-758, equals the
  pairwise f_iou).  box_loss_fn / segm_loss_fn in {'iou', 'wt_cov'}."""
  dt = fwd['y_out'].dtype
  y_gt, s_gt = y_gt.astype(dt), s_gt.astype(dt)
  B, T = s_gt.shape
  y_out, s_out, attn_box = fwd['y_out'], fwd['s_out'], fwd['attn_box']
  fixed_order = bool(_opt(opt, 'fixed_order', False))
  # get_gt_attn -> get_gt_box, full_model.py:561-566 (min_padding = padding + 4)
  _, _, attn_box_gt = get_gt_box(y_gt, padding_ratio=opt['attn_box_padding_ratio'],
                                 center_shift_ratio=0.0, min_padding=opt['padding'] + 4.0)
  out = {'attn_box_gt': attn_box_gt}
  identity = get_identity_match(s_gt)
  iou_box_pair = f_iou_pairwise(attn_box, attn_box_gt)
  if fixed_order:
    match_box = identity
    iou_box_mask = np.einsum('bii->bi', iou_box_pair)  # f_iou(pairwise=False), :924
  else:
    match_box = f_segm_match(iou_box_pair, s_gt)
    iou_box_mask = (iou_box_pair * match_box).sum(axis=1)
  out['match_box'] = match_box
  cnt_box = np.maximum(1.0, match_box.sum(axis=(1, 2)))
  iou_soft_box = (iou_box_mask.sum(axis=1) / cnt_box).sum() / B
  box_fn = _opt(opt, 'box_loss_fn', 'iou')
  if box_fn == 'iou':
    box_loss = -iou_soft_box
  elif box_fn == 'wt_cov':  # :967-968 passes the SCALAR iou_soft_box; not restatable -> pairwise
    raise NotImplementedError('box_loss_fn wt_cov feeds a scalar to f_weighted_coverage (:968)')
  else:
    raise NotImplementedError(box_fn)
  out['iou_soft_box'], out['box_loss'] = iou_soft_box, box_loss

  iou_pair = f_iou_pairwise(y_out, y_gt)
  real_match = f_segm_match(iou_pair, s_gt)
  match = identity if fixed_order else real_match
  out['match'] = match
  cnt = np.maximum(1.0, match.sum(axis=(1, 2)))
  out['wt_cov_soft'] = f_weighted_coverage(iou_pair, y_gt)
  out['unwt_cov_soft'] = f_unweighted_coverage(iou_pair, cnt)
  if fixed_order:
    iou_mask = np.einsum('bii->bi', iou_pair)
  else:
    iou_mask = (iou_pair * match).sum(axis=1)
  out['iou_soft'] = (iou_mask.sum(axis=1) / cnt).sum() / B
  segm_fn = _opt(opt, 'segm_loss_fn', 'iou')
  if segm_fn == 'iou':
    segm_loss = -out['iou_soft']
  elif segm_fn == 'wt_cov':
    segm_loss = -out['wt_cov_soft']
  else:
    raise NotImplementedError(segm_fn)
  out['segm_loss'] = segm_loss
  out['conf_loss'] = f_conf_loss(s_out, match)
  out['loss'] = box_loss + segm_loss + _opt(opt, 'loss_mix_ratio', 1.0) * out['conf_loss']

  y_hard = (y_out > 0.5).astype(dt)
  iou_hard = f_iou_pairwise(y_hard, y_gt)
  out['wt_cov_hard'] = f_weighted_coverage(iou_hard, y_gt)
  out['unwt_cov_hard'] = f_unweighted_coverage(iou_hard, cnt)
  out['iou_hard'] = ((iou_hard * real_match).sum(axis=(1, 2)) / cnt).sum() / B
  dice = f_dice_pairwise(y_hard, y_gt)
  out['dice'] = ((dice * real_match).sum(axis=(1, 2)) / cnt).sum() / B
  out['count_acc'], out['dic'], out['dic_abs'] = f_count_stats(s_out, s_gt)
  out['iou_soft_pairwise'], out['iou_hard_pairwise'], out['dice_pairwise'] = iou_pair, iou_hard, dice
  return out


# --------------------------------------------------------------------------------------
# evaluation post-processing (utils/postprocess.py) and metrics (analysis.py:314-787)
# --------------------------------------------------------------------------------------
def pp_apply_confidence(y_out, s_out):
  """postprocess.py:15-29."""
  return y_out * s_out[:, :, None, None], (s_out > 0.5).astype('float')


def pp_apply_one_label(y_out):
  """postprocess.py:32-52 (numpy argmax: first maximum)."""
  out = np.zeros(y_out.shape)
  for ii in range(y_out.shape[0]):
    am = np.argmax(y_out[ii], axis=0)
    for jj in range(y_out.shape[1]):
      out[ii, jj] = (am == jj).astype('float32') * y_out[ii, jj]
  return out


def pp_apply_threshold(y_out, thresh):
  """postprocess.py:5-12."""
  return (y_out > thresh).astype('float32')


def pp_morph(y_out):
  """postprocess.py:55-72: cv2.dilate(plane, ones((5, 5))) — the maximum over the 5 x 5 window, pixels outside the image
  ignored (cv2's default border value for a dilation).  cv2 itself is not part of this stack: restated from its documented
  behaviour."""
  H, W = y_out.shape[-2:]
  pad = np.pad(y_out, [(0, 0)] * (y_out.ndim - 2) + [(2, 2), (2, 2)], constant_values=-np.inf)
  out = np.full(y_out.shape, -np.inf)
  for dy in range(5):
    for dx in range(5):
      out = np.maximum(out, pad[..., dy:dy + H, dx:dx + W])
  return out


def pp_upsample(y_out, H, W, sigma_color=10.0, sigma_space=10.0):
  """postprocess.py:75-106: cv2.resize(a, (W, H), INTER_LINEAR) (pixel centres aligned: source coordinate
  (d + 0.5) * src / dst - 0.5 clamped to the image) then cv2.bilateralFilter(b, 5, 10, 10) (circular radius-2 neighbourhood,
  BORDER_REFLECT_101; the exact exponentials).  Unpinned: cv2 is not part of this stack."""
  Hs, Ws = y_out.shape[-2:]

  def taps(n_src, n_dst):
    f = (np.arange(n_dst) + 0.5) * (n_src / n_dst) - 0.5
    i0 = np.floor(f).astype(int)
    w = f - i0
    lo = i0 < 0
    hi = i0 >= n_src - 1
    i0 = np.clip(i0, 0, n_src - 1)
    w = np.where(lo | hi, 0.0, w)
    return i0, np.minimum(i0 + 1, n_src - 1), w
  y0, y1, wy = taps(Hs, H)
  x0, x1, wx = taps(Ws, W)
  a = y_out.astype(np.float64)
  top = a[..., y0, :][..., :, x0] * (1 - wx) + a[..., y0, :][..., :, x1] * wx
  bot = a[..., y1, :][..., :, x0] * (1 - wx) + a[..., y1, :][..., :, x1] * wx
  b = top * (1 - wy)[:, None] + bot * wy[:, None]
  ref = lambda i, n: np.abs(i) if n == 1 else (lambda j: np.where(j >= n, 2 * (n - 1) - j, j))(np.abs(i))
  num, den = np.zeros_like(b), np.zeros_like(b)
  rr, cc = np.arange(H), np.arange(W)
  for dy in range(-2, 3):
    for dx in range(-2, 3):
      if dy * dy + dx * dx > 4:
        continue
      v = b[..., ref(rr + dy, H), :][..., :, ref(cc + dx, W)]
      wgt = np.exp(-(dy * dy + dx * dx) / (2 * sigma_space ** 2) - (v - b) ** 2 / (2 * sigma_color ** 2))
      num += wgt * v
      den += wgt
  return num / den


def pp_remove_tiny(y_out, conf, threshold=200):
  """postprocess.py:109-136."""
  if threshold == 0:
    return y_out, conf
  size = y_out.sum(axis=(2, 3), keepdims=True)
  keep = (size > threshold).astype('float32')
  return y_out * keep, conf * keep.reshape(conf.shape)


def postprocess(y_out, s_out, thresh, fg=None, remove_tiny_threshold=0):
  """full_model_eval.py:112-124 without upsample / morph (cv2)."""
  y, s = pp_apply_confidence(y_out, s_out)
  y = pp_apply_threshold(pp_apply_one_label(y), thresh)
  if fg is not None:
    y = y * fg[:, None]  # mask_foreground, postprocess.py:139-147
    y, s = pp_remove_tiny(y, s, remove_tiny_threshold)
  return y, s


def an_f_iou(a, b):
  """analysis.py:314-326."""
  inter = (a * b).sum(axis=-1).sum(axis=-1)
  union = (a + b).sum(axis=-1).sum(axis=-1) - inter
  return inter / (union + np.equal(union, 0).astype('float32'))


def an_f_iou_pairwise(a, b):
  """analysis.py:329-334 (one example: a [N,H,W], b [M,H,W])."""
  return an_f_iou(np.expand_dims(a, 1), np.expand_dims(b, 0))


def _an_f_pr(a, b):
  """analysis.py:337-349."""
  inter = (a * b).sum(axis=-1).sum(axis=-1)
  asum = a.sum(axis=-1).sum(axis=-1)
  return inter / (asum + np.equal(asum, 0).astype('float32'))


def _an_f_dice(a, b):
  """analysis.py:352-367."""
  ca, cb = a.sum(axis=-1).sum(axis=-1), b.sum(axis=-1).sum(axis=-1)
  cs = ca + cb
  return 2 * (a * b).sum(axis=-1).sum(axis=-1) / (cs + np.equal(cs, 0).astype('float32'))


def eval_metrics(y_out, y_gt, s_gt):
  """analysis.py:370-787 per image, for binary y_out / y_gt [B,T,H,W]: dict of [B] arrays and
  the variable-length per-instance lists (avg_pr, avg_re, obj_pr, obj_re)."""
  B, T = s_gt.shape
  num_obj = np.maximum(s_gt.sum(axis=1), 1)           # :773-787
  count_out = (y_out.sum(axis=(2, 3)) > 0).astype('float32')  # :766-770
  count_gt = s_gt.sum(axis=1)
  out = {k: np.zeros(B) for k in ('sbd', 'wt_cov', 'unwt_cov', 'fg_iou', 'fg_dice', 'avg_fp', 'avg_fn')}
  lists = {k: [] for k in ('avg_pr', 'avg_re', 'obj_pr', 'obj_re')}
  ious = []
  for ii in range(B):
    a, b, no = y_out[ii], y_gt[ii], int(num_obj[ii])
    iou = an_f_iou_pairwise(a, b)
    ious.append(iou)
    bd_a = np.array([_an_f_dice(a[k:k + 1], b).max(axis=0) for k in range(T)])   # :370-386
    bd_b = np.array([_an_f_dice(b[k:k + 1], a).max(axis=0) for k in range(T)])
    out['sbd'][ii] = min(bd_a[:no].mean(), bd_b[:no].mean())                       # :434-460
    cov = iou.max(axis=0)                                                          # :463-464
    tot = b.sum()
    w_wt = b.sum(axis=(1, 2)) / (tot + np.equal(tot, 0).astype('float32'))         # :467-478
    out['wt_cov'][ii] = (cov * w_wt)[:no].sum()
    out['unwt_cov'][ii] = (cov * (1 / num_obj[ii]))[:no].sum()
    out['fg_iou'][ii] = an_f_iou(a.max(axis=0), b.max(axis=0))                     # :533-553
    out['fg_dice'][ii] = _an_f_dice(a.max(axis=0), b.max(axis=0))                  # :556-576
    out['avg_fp'][ii] = (count_out[ii] * np.equal(iou.sum(axis=1), 0)).sum()       # :579-592
    out['avg_fn'][ii] = (s_gt[ii] * np.equal(iou.sum(axis=0), 0)).sum()            # :595-605
    pr = _an_f_pr(a, b.max(axis=0, keepdims=True))                                 # :608-627
    re = _an_f_pr(b, a.max(axis=0, keepdims=True))                                 # :630-650
    m_out = (iou.max(axis=1) >= 0.5).astype('float32')                             # :653-671
    m_gt = (iou.max(axis=0) >= 0.5).astype('float32')                              # :674-690
    for jj in range(T):
      if count_out[ii, jj] > 0:
        lists['avg_pr'].append(pr[jj])
        lists['obj_pr'].append(m_out[jj])
    for jj in range(int(count_gt[ii])):
      lists['avg_re'].append(re[jj])
      lists['obj_re'].append(m_gt[jj])
  d = count_out.sum(axis=1) - count_gt
  out.update(count_mse=d.astype('float') ** 2, count_acc=(d == 0).astype('float'), dic=d,   # :693-763
             dic_abs=np.abs(d), iou_pairwise=np.array(ious))
  out.update({k: np.array(v) for k, v in lists.items()})
  return out


def random_transformation(x, padding, off_y, off_x, flip_v=False, flip_h=False, transpose=False):
  """image_ops.py:9-113 with phase_train true, for GIVEN draws, on x [N,H,W,C] (use C = 1 planes
  for the instance masks): zero-pad (:37), crop at the offset (:53-55), tf.reverse along H / W
  (:88-91), tf.transpose of H and W (:95-97)."""
  N, H, W = x.shape[:3]
  pad = [(0, 0), (padding, padding), (padding, padding)] + [(0, 0)] * (x.ndim - 3)
  r = np.pad(x, pad)[:, off_y:off_y + H, off_x:off_x + W]
  if flip_v:
    r = r[:, ::-1]
  if flip_h:
    r = r[:, :, ::-1]
  if transpose:
    r = np.swapaxes(r, 1, 2)
  return np.ascontiguousarray(r)


def colour_jitter(x, hue, saturation, brightness, contrast):
  """image_ops.py:99-103 for GIVEN draws on x [B,H,W,3]: random_hue (:116-147, tf.image.adjust_hue: RGB -> HSV, hue =
  (hue + delta + 1) mod 1, back), random_saturation (:150-180, adjust_saturation: saturation * factor clipped to [0, 1]),
  tf.image.random_brightness (x + delta; float images are not clipped) and tf.image.random_contrast ((x - mean) * factor +
  mean, mean per image and channel over H x W).  The colour-space conversions follow TensorFlow's published kernels
  (core/kernels/colorspace_op.h): they live in TF-0.12, which cannot run here — unpinned like tf.round."""
  x = np.asarray(x, dtype=np.float64)
  r, g, b = x[..., 0], x[..., 1], x[..., 2]
  v = np.maximum(r, np.maximum(g, b))
  rng_ = v - np.minimum(r, np.minimum(g, b))
  s = np.where(v > 0, rng_ / np.where(v > 0, v, 1.0), 0.0)
  with np.errstate(divide='ignore', invalid='ignore'):
    norm = 1.0 / (6.0 * rng_)
    h = np.where(r == v, norm * (g - b), np.where(g == v, norm * (b - r) + 2.0 / 6.0, norm * (r - g) + 4.0 / 6.0))
  h = np.where(rng_ > 0, h, 0.0)
  h = np.where(h < 0, h + 1.0, h)
  h = np.mod(h + (hue + 1.0), 1.0)
  s = np.clip(s * saturation, 0.0, 1.0)
  d6 = h * 6.0
  dr = np.clip(np.abs(d6 - 3.0) - 1.0, 0.0, 1.0)
  dg = np.clip(2.0 - np.abs(d6 - 2.0), 0.0, 1.0)
  db = np.clip(2.0 - np.abs(d6 - 4.0), 0.0, 1.0)
  out = np.stack([(1.0 - s + s * dr) * v, (1.0 - s + s * dg) * v, (1.0 - s + s * db) * v], axis=-1) + brightness
  mean = out.mean(axis=(1, 2), keepdims=True)
  return (out - mean) * contrast + mean


# --------------------------------------------------------------------------------------
# model option handling shared by both graphs
# --------------------------------------------------------------------------------------


def _opt(opt, key, default):
  return opt[key] if key in opt else default


def derive(opt, box_model=False):
  """The shape/flag bookkeeping of full_model.py:18-160,239-258,305-313,455-459,494-502
  (box_model.py:16-82,343-352 when box_model=True)."""
  d = {}
  d['T'] = opt['timespan']
  d['H'] = opt['inp_height']
  d['W'] = opt['inp_width']
  d['D'] = opt['inp_depth']
  d['Fh'] = opt['filter_height']
  d['Fw'] = opt['filter_width']
  add_d = _opt(opt, 'add_d_out', False)
  add_y = _opt(opt, 'add_y_out', False) if 'add_d_out' in opt else False
  nsc = _opt(opt, 'num_semantic_classes', 1)
  d['add_d_out'], d['add_y_out'], d['nsc'] = add_d, add_y, nsc
  if box_model:
    c_inp, c_can, c_d, c_y = True, True, add_d, add_y
    a_inp, a_can, a_d, a_y = c_inp, c_can, c_d, c_y
  else:
    if 'attn_add_d_out' in opt:
      a_d, a_y = opt['attn_add_d_out'], opt['attn_add_y_out']
      a_inp, a_can = opt['attn_add_inp'], opt['attn_add_canvas']
    else:
      a_d, a_y, a_inp, a_can = add_d, add_y, True, True
    if 'ctrl_add_d_out' in opt:
      c_d, c_y = opt['ctrl_add_d_out'], opt['ctrl_add_y_out']
      c_inp, c_can = opt['ctrl_add_inp'], opt['ctrl_add_canvas']
    else:
      c_d, c_y = add_d, add_y
      c_inp = c_can = not c_d
  d['ctrl_in'] = (c_inp, c_can, c_d, c_y)
  d['attn_in'] = (a_inp, a_can, a_d, a_y)
  depth = lambda f: (d['D'] if f[0] else 0) + (1 if f[1] else 0) + (8 if f[2] else 0) + \
      (nsc if f[3] else 0)
  d['ccnn_inp_depth'] = depth(d['ctrl_in'])
  d['acnn_inp_depth'] = depth(d['attn_in'])
  d['ccnn_nlayers'] = len(opt['ctrl_cnn_filter_size'])
  d['ccnn_channels'] = [d['ccnn_inp_depth']] + list(opt['ctrl_cnn_depth'])
  d['ccnn_pool'] = list(opt['ctrl_cnn_pool'])
  sub = int(np.prod(d['ccnn_pool']))
  d['gh'], d['gw'] = d['H'] // sub, d['W'] // sub  # Python-2 int division
  d['G'] = d['gh'] * d['gw']
  d['hid'] = opt['ctrl_rnn_hid_dim']
  d['iters'] = opt['num_ctrl_rnn_iter']
  d['n_gmlp'] = opt['num_glimpse_mlp_layers']
  d['n_cmlp'] = opt['num_ctrl_mlp_layers']
  d['squash'] = opt['squash_ctrl_params']
  d['fixed_var'] = _opt(opt, 'fixed_var', True if box_model else False)
  d['dynamic_var'] = _opt(opt, 'dynamic_var', False)
  d['use_bn'] = opt['use_bn']
  if box_model:
    return d
  d['fixed_gamma'] = opt['fixed_gamma']
  d['disable_overwrite'] = _opt(opt, 'disable_overwrite', True)
  d['acnn_nlayers'] = len(opt['attn_cnn_filter_size'])
  d['acnn_channels'] = [d['acnn_inp_depth']] + list(opt['attn_cnn_depth'])
  d['acnn_pool'] = list(opt['attn_cnn_pool'])
  asub = int(np.prod(d['acnn_pool']))
  d['core_depth'] = d['acnn_channels'][-1]
  d['core_dim'] = (d['Fh'] // asub) * (d['Fw'] // asub) * d['core_depth']
  d['adcnn_nlayers'] = len(opt['attn_dcnn_filter_size'])
  d['adcnn_unpool'] = list(opt['attn_dcnn_pool'])
  d['adcnn_channels'] = [d['core_depth']] + list(opt['attn_dcnn_depth'])
  add_skip = _opt(opt, 'add_skip_conn', True)
  d['add_skip_conn'] = add_skip
  skip_flags = _opt(opt, 'attn_cnn_skip', [add_skip] * d['acnn_nlayers'])
  d['skip_rev'] = list(skip_flags[::-1])  # may be a str: every char truthy (§5 quirk)
  if add_skip:
    ch_rev = d['acnn_channels'][::-1][1:] + [d['acnn_inp_depth']]
    d['skip_ch'] = [0] + [ch if sk else 0 for sk, ch in zip(d['skip_rev'], ch_rev)]
  else:
    d['skip_ch'] = None
  return d


def param_shapes(opt, box_model=False):
  """Names/shapes of every weight the path registers into ``model`` (nnlib.py:120-127,
  206-211,333-335,471-474,611-623), in the reference's key scheme."""
  d = derive(opt, box_model)
  S = {}
  T = d['T']

  def cnn(scope, ch, nl):
    for i in range(nl):
      S['%s_w_%d' % (scope, i)] = (3, 3, ch[i], ch[i + 1])
      S['%s_b_%d' % (scope, i)] = (ch[i + 1],)
      for t in range(T):
        for n in ('beta', 'gamma', 'ema_mean', 'ema_var'):
          S['%s_%d_%d_%s' % (scope, i, t, n)] = (ch[i + 1],)

  cnn('ctrl_cnn', d['ccnn_channels'], d['ccnn_nlayers'])
  Cf, hid = d['ccnn_channels'][-1], d['hid']
  for g in 'ifuo':
    S['ctrl_lstm_w_x' + g] = (Cf, hid)
    S['ctrl_lstm_w_h' + g] = (hid, hid)
    S['ctrl_lstm_b_' + g] = (hid,)
  gdims = [hid] * d['n_gmlp'] + [d['G']]
  for i in range(d['n_gmlp']):
    S['glimpse_mlp_w_%d' % i] = (gdims[i], gdims[i + 1])
    S['glimpse_mlp_b_%d' % i] = (gdims[i + 1],)
  cdims = [hid] + [opt['ctrl_mlp_dim']] * (d['n_cmlp'] - 1) + [9]
  for i in range(d['n_cmlp']):
    S['ctrl_mlp_w_%d' % i] = (cdims[i], cdims[i + 1])
    S['ctrl_mlp_b_%d' % i] = (cdims[i + 1],)
  if box_model:
    S['score_mlp_w_0'] = (hid, d['nsc'])
    S['score_mlp_b_0'] = (d['nsc'],)
    return S
  cnn('attn_cnn', d['acnn_channels'], d['acnn_nlayers'])
  in_ch = d['adcnn_channels'][0]
  for i in range(d['adcnn_nlayers']):
    out_ch = d['adcnn_channels'][i + 1]
    if d['skip_ch'] is not None:
      in_ch += d['skip_ch'][i]
    S['attn_dcnn_w_%d' % i] = (3, 3, out_ch, in_ch)
    S['attn_dcnn_b_%d' % i] = (out_ch,)
    for t in range(T):
      for n in ('beta', 'gamma', 'ema_mean', 'ema_var'):
        S['attn_dcnn_%d_%d_%s' % (i, t, n)] = (out_ch,)
    in_ch = out_ch
  S['score_mlp_w_0'] = (hid + d['core_dim'], 1)
  S['score_mlp_b_0'] = (1,)
  return S


def random_params(opt, seed, box_model=False, w_scale=None):
  """Seeded NON-reference initialisation used by fixtures/tests: fan-in scaled normals and
  non-trivial BN statistics so masks are not all sigma(-5)."""
  rng = np.random.RandomState(seed)
  P = {}
  for k, shp in sorted(param_shapes(opt, box_model).items()):
    if k.endswith('_beta'):
      v = rng.normal(0.1, 0.2, shp)
    elif k.endswith('_gamma'):
      v = rng.uniform(0.7, 1.4, shp)
    elif k.endswith('_ema_mean'):
      v = rng.normal(0.0, 0.2, shp)
    elif k.endswith('_ema_var'):
      v = rng.uniform(0.5, 1.5, shp)
    elif len(shp) == 1:
      v = rng.normal(0.0, 0.1, shp)
      if k == 'ctrl_lstm_b_f':
        v = v + 1.0
    else:
      fan_in = int(np.prod(shp[:-1])) if len(shp) == 2 else 9 * (
          shp[3] if 'dcnn' in k else shp[2])
      s = w_scale if w_scale is not None else 1.3 / np.sqrt(fan_in)
      v = rng.normal(0.0, s, shp)
    P[k] = v.astype(np.float32)
  d_ = derive(opt, box_model)
  if not box_model:  # make the decoder's last layer swing, so masks are not all sigma(-5)
    last = d_['adcnn_nlayers'] - 1
    for t in range(d_['T']):
      P['attn_dcnn_%d_%d_beta' % (last, t)] = rng.normal(0.4, 0.2, (1,)).astype(np.float32)
      P['attn_dcnn_%d_%d_gamma' % (last, t)] = rng.uniform(1.5, 3.0, (1,)).astype(np.float32)
  # keep the predicted box inside the image and a sensible size
  P['ctrl_mlp_w_%d' % (derive(opt, box_model)['n_cmlp'] - 1)] *= 0.5
  b = P['ctrl_mlp_b_%d' % (derive(opt, box_model)['n_cmlp'] - 1)]
  b[2:4] = np.log(0.35)
  b[4:6] = np.log(2.0)
  b[6:9] = [0.1, 0.3, 1.2]
  return P


# --------------------------------------------------------------------------------------
# the decode loop
# --------------------------------------------------------------------------------------


def _controller(d, P, feat, dt):
  """full_model.py:668-689 (= box_model.py:416-444): glimpse read-out + LSTM + gMLP."""
  B = feat.shape[0]
  hid, G = d['hid'], d['G']
  state = np.zeros((B, 2 * hid), dtype=dt)
  gmap = np.ones((B, G, 1), dtype=dt) / G
  gmaps = []
  gacts = [relu] * (d['n_gmlp'] - 1) + [softmax]
  for it in range(d['iters']):
    gmaps.append(gmap[:, :, 0])
    glimpse = (feat * gmap).sum(axis=1)
    state, _, _, _ = lstm_step(glimpse, state, P, 'ctrl_lstm', hid)
    h = state[:, hid:]
    hg = run_mlp(h, P, 'glimpse_mlp', gacts)
    if it < d['iters'] - 1:
      gmap = hg[-1][:, :, None]
  cacts = [relu] * (d['n_cmlp'] - 1) + [None]
  ctrl_out = run_mlp(h, P, 'ctrl_mlp', cacts)[-1]
  return h, ctrl_out, np.stack(gmaps, axis=1)


def _decode_ctrl(d, ctrl_out, dt):
  """full_model.py:691-722 / box_model.py:446-468."""
  ctr_norm = ctrl_out[:, 0:2]
  lg_size = ctrl_out[:, 2:4]
  if d['squash']:
    ctr_norm = np.tanh(ctr_norm)
    lg_size = -np.log1p(np.exp(lg_size))
  ctr, size = get_unnormalized_attn(ctr_norm, lg_size, d['H'], d['W'])
  if d['fixed_var']:
    lg_var = np.zeros_like(ctr)
  else:
    lg_var = get_normalized_var(size, d['Fh'], d['Fw'])
  if d['dynamic_var']:
    lg_var = ctrl_out[:, 4:6]
  return ctr_norm, lg_size, ctr, size, lg_var


def _cat_inputs(flags, x, canvas, d_in, y_in):
  """full_model.py:640-661: order x, canvas, d_in, y_in."""
  parts = []
  if flags[0]:
    parts.append(x)
  if flags[1]:
    parts.append(canvas)
  if flags[2]:
    parts.append(d_in)
  if flags[3]:
    parts.append(y_in)
  return np.concatenate(parts, axis=3)


def full_model_forward(opt, P, x, d_in=None, y_in=None, dtype=np.float64):
  """Eval-mode (phase_train=False, use_knob=False) forward of full_model.py:638-907.

  Returns a dict with the reference's output keys (y_out [B,T,H,W], s_out [B,T],
  x_patch, y_out_patch, attn_box, attn_ctr, attn_size, attn_lg_var, ctrl_out,
  ctrl_rnn_glimpse_map, h_core, canvas)."""
  dt = np.dtype(dtype)
  d = derive(opt)
  P = {k: v.astype(dt) for k, v in P.items()}
  x = x.astype(dt)
  d_in = None if d_in is None else d_in.astype(dt)
  y_in = None if y_in is None else y_in.astype(dt)
  B, T, H, W, Fh, Fw = x.shape[0], d['T'], d['H'], d['W'], d['Fh'], d['Fw']
  canvas = np.zeros((B, H, W, 1), dtype=dt)
  out = {k: [] for k in ('y_out', 's_out', 'x_patch', 'y_out_patch', 'attn_box', 'attn_ctr',
                         'attn_size', 'attn_lg_var', 'ctrl_out', 'ctrl_rnn_glimpse_map',
                         'h_core', 'h_crnn')}
  for tt in range(T):
    ccnn_inp = _cat_inputs(d['ctrl_in'], x, canvas, d_in, y_in)
    acnn_inp = _cat_inputs(d['attn_in'], x, canvas, d_in, y_in)
    h_ccnn = run_cnn(ccnn_inp, P, 'ctrl_cnn', d['ccnn_nlayers'], d['ccnn_pool'], tt,
                     d['use_bn'])
    feat = h_ccnn[-1].reshape(B, d['G'], d['ccnn_channels'][-1])
    h, ctrl_out, gmaps = _controller(d, P, feat, dt)
    _, _, ctr, size, lg_var = _decode_ctrl(d, ctrl_out, dt)
    if d['fixed_gamma']:
      attn_gamma = np.ones((B, 1, 1, 1), dtype=dt)  # exp(0.0), full_model.py:712,719
      y_lg_gamma = np.full((B, 1, 1, 1), 2.0, dtype=dt)
    else:
      attn_gamma = np.exp(ctrl_out[:, 6:7]).reshape(-1, 1, 1, 1)
      y_lg_gamma = ctrl_out[:, 8:9].reshape(-1, 1, 1, 1)
    box_gamma = np.exp(ctrl_out[:, 7:8]).reshape(-1, 1, 1, 1)
    f_y = get_gaussian_filter(ctr[:, 0], size[:, 0], lg_var[:, 0], H, Fh)
    f_x = get_gaussian_filter(ctr[:, 1], size[:, 1], lg_var[:, 1], W, Fw)
    f_y_inv = np.transpose(f_y, (0, 2, 1))
    f_x_inv = np.transpose(f_x, (0, 2, 1))
    ones = np.ones((B, Fh, Fw, 1), dtype=dt)
    attn_box = sigmoid(extract_patch(ones * box_gamma, f_y_inv, f_x_inv, 1) - 5.0)
    attn_box = attn_box.reshape(B, 1, H, W)
    x_patch = attn_gamma * extract_patch(acnn_inp, f_y, f_x, d['acnn_inp_depth'])
    h_acnn = run_cnn(x_patch, P, 'attn_cnn', d['acnn_nlayers'], d['acnn_pool'], tt,
                     d['use_bn'])
    h_core = h_acnn[-1].reshape(B, d['core_dim'])
    if d['add_skip_conn']:
      h_rev = h_acnn[::-1][1:] + [x_patch]
      skip = [None] + [hh if sk else None for sk, hh in zip(d['skip_rev'], h_rev)]
    else:
      skip = None
    h_adcnn = run_dcnn(h_acnn[-1], P, 'attn_dcnn', d['adcnn_nlayers'], d['adcnn_unpool'],
                       tt, skip, d['use_bn'])
    y = extract_patch(h_adcnn[-1], f_y_inv, f_x_inv, 1)
    y = sigmoid(np.exp(y_lg_gamma) * y - 5.0).reshape(B, 1, H, W)
    if d['disable_overwrite']:
      y = (1 - canvas).reshape(B, 1, H, W) * y
    s = run_mlp(np.concatenate([h, h_core], axis=1), P, 'score_mlp', [sigmoid])[-1]
    canvas = np.maximum(y.reshape(B, H, W, 1), canvas)
    for k, v in (('y_out', y), ('s_out', s), ('x_patch', x_patch[:, None]),
                 ('y_out_patch', h_adcnn[-1][:, None]), ('attn_box', attn_box),
                 ('attn_ctr', ctr[:, None]), ('attn_size', size[:, None]),
                 ('attn_lg_var', lg_var[:, None]), ('ctrl_out', ctrl_out[:, None]),
                 ('ctrl_rnn_glimpse_map', gmaps[:, None]), ('h_core', h_core[:, None]),
                 ('h_crnn', h[:, None])):
      out[k].append(v)
  res = {k: np.concatenate(v, axis=1) for k, v in out.items()}
  res['ctrl_rnn_glimpse_map'] = res['ctrl_rnn_glimpse_map'].reshape(
      B, T, d['iters'], d['gh'], d['gw'])
  res['canvas'] = canvas
  return res


def box_model_forward(opt, P, x, y_gt, noise, d_in=None, y_in=None, dtype=np.float64):
  """Forward of box_model.py:403-513 (fixed_order=False, use_iou_box=False path).

  ``noise`` [T,B,H,W,1] stands in for the graph's tf.random_uniform(0, 0.3) draws
  (box_model.py:500-502) so the computation is reproducible."""
  dt = np.dtype(dtype)
  d = derive(opt, box_model=True)
  P = {k: v.astype(dt) for k, v in P.items()}
  x, y_gt, noise = x.astype(dt), y_gt.astype(dt), noise.astype(dt)
  B, T, H, W, Fh, Fw = x.shape[0], d['T'], d['H'], d['W'], d['Fh'], d['Fw']
  canvas = np.zeros((B, H, W, 1), dtype=dt)
  _, _, attn_box_gt = get_gt_box(
      y_gt, padding_ratio=opt['attn_box_padding_ratio'], center_shift_ratio=0.0)
  matched = np.zeros((B, T), dtype=dt)  # grd_match_cum is never updated (box_model.py:398)
  out = {k: [] for k in ('s_out', 'attn_box', 'attn_ctr', 'attn_size', 'ctrl_out')}
  for tt in range(T):
    ccnn_inp = _cat_inputs(d['ctrl_in'], x, canvas,
                           None if d_in is None else d_in.astype(dt),
                           None if y_in is None else y_in.astype(dt))
    h_ccnn = run_cnn(ccnn_inp, P, 'ctrl_cnn', d['ccnn_nlayers'], d['ccnn_pool'], tt,
                     d['use_bn'])
    feat = h_ccnn[-1].reshape(B, d['G'], d['ccnn_channels'][-1])
    h, ctrl_out, _ = _controller(d, P, feat, dt)
    _, _, ctr, size, lg_var = _decode_ctrl(d, ctrl_out, dt)
    box_gamma = np.exp(ctrl_out[:, 7:8]).reshape(-1, 1, 1, 1)
    f_y = get_gaussian_filter(ctr[:, 0], size[:, 0], lg_var[:, 0], H, Fh)
    f_x = get_gaussian_filter(ctr[:, 1], size[:, 1], lg_var[:, 1], W, Fw)
    ones = np.ones((B, Fh, Fw, 1), dtype=dt)
    attn_box = box_gamma * extract_patch(ones, np.transpose(f_y, (0, 2, 1)),
                                         np.transpose(f_x, (0, 2, 1)), 1)
    attn_box = sigmoid(attn_box - 5.0).reshape(B, 1, H, W)
    iou = f_inter(attn_box, attn_box_gt) / f_union(attn_box, attn_box_gt, eps=1e-5)
    gm = f_greedy_match(iou, matched)[:, :, None, None]
    y_sel = (gm * y_gt).sum(axis=1)[..., None]
    y_sel = y_sel - y_sel * noise[tt]
    canvas = np.maximum(y_sel, canvas)
    s = run_mlp(h, P, 'score_mlp', [None])[-1]
    s = sigmoid(s) if d['nsc'] == 1 else softmax(s)
    for k, v in (('s_out', s[:, None]), ('attn_box', attn_box), ('attn_ctr', ctr[:, None]),
                 ('attn_size', size[:, None]), ('ctrl_out', ctrl_out[:, None])):
      out[k].append(v)
  res = {k: np.concatenate(v, axis=1) for k, v in out.items()}
  if d['nsc'] == 1:
    res['s_out'] = res['s_out'][:, :, 0]
  res['canvas'] = canvas
  return res


# --------------------------------------------------------------------------------------
# configurations named in SURVEY.md §8 (run_cvppp.sh:37-72, run_kitti.sh:68-111,
# run_cityscapes.sh:62-110 -> full_model_train.py:581-658 make_opt)
# --------------------------------------------------------------------------------------


def make_opt(arch, H, W, T, **over):
  base = dict(
      inp_height=H, inp_width=W, inp_depth=3, padding=16, filter_height=48, filter_width=48,
      timespan=T, ctrl_rnn_hid_dim=256, num_ctrl_mlp_layers=1, ctrl_mlp_dim=256,
      mlp_dropout=None, weight_decay=5e-5, use_bn=True, attn_box_padding_ratio=0.2,
      use_knob=False, squash_ctrl_params=False, fixed_order=False, fixed_gamma=False,
      fixed_var=False, dynamic_var=False, num_ctrl_rnn_iter=5, num_glimpse_mlp_layers=2,
      stop_canvas_grad=True, use_iou_box=False, add_skip_conn=False, disable_overwrite=False,
      add_d_out=False, add_y_out=False, num_semantic_classes=1, ctrl_add_inp=True,
      ctrl_add_canvas=True, ctrl_add_d_out=False, ctrl_add_y_out=False, attn_add_inp=True,
      attn_add_canvas=True, attn_add_d_out=False, attn_add_y_out=False,
      ctrl_cnn_filter_size=[3] * 8, attn_cnn_filter_size=[3] * 6,
      attn_dcnn_filter_size=[3] * 7, attn_dcnn_pool=[2, 1, 2, 1, 2, 1, 1],
      attn_cnn_pool=[1, 2, 1, 2, 1, 2])
  if arch == 'cvppp':
    base.update(ctrl_cnn_depth=[8, 8, 16, 16, 32, 32, 64, 64],
                ctrl_cnn_pool=[1, 2, 1, 2, 1, 2, 2, 2],
                attn_cnn_depth=[8, 8, 16, 16, 32, 32],
                attn_dcnn_depth=[32, 32, 16, 16, 8, 8, 1], fixed_gamma=True,
                attn_cnn_skip='1,1,1')
  elif arch in ('kitti', 'cityscapes'):
    base.update(ctrl_cnn_depth=[16, 16, 32, 32, 64, 64, 64, 64],
                ctrl_cnn_pool=[2, 2, 1, 2, 1, 2, 1, 2],
                attn_cnn_depth=[16, 32, 32, 64, 64, 96],
                attn_dcnn_depth=[64, 64, 32, 32, 16, 16, 1], dynamic_var=True,
                add_skip_conn=True, add_d_out=True, add_y_out=True, attn_add_d_out=True,
                attn_add_y_out=True, ctrl_add_d_out=True, ctrl_add_y_out=True,
                attn_cnn_skip='1,0,1,0,1,0,1,0')
    if arch == 'cityscapes':
      base.update(num_semantic_classes=9, fixed_gamma=True, use_iou_box=True)
  else:
    raise ValueError(arch)
  base.update(over)
  return base
