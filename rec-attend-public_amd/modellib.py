"""`modellib` surface of the reference on gfx950 kernels (cited as modellib.py:line).

Built: the Gaussian-attention operators of the decode loop (get_gaussian_filter,
extract_patch and the (un)normalisers), the Hungarian op binding with f_segm_match's
conditioning around it, and the loss / IoU family (f_iou, f_dice, f_inter, f_union, coverage,
f_conf_loss, f_greedy_match, get_gt_box, ...; modellib.py:28-531) on the streaming kernels of
csrc/ra_loss.hip.
"""
import math

import numpy as np
import torch

import ra_ops as ops
from ra_native import RecAttendError

hungarian = ops.hungarian  # hungarian_module.hungarian (modellib.py:389-393,406)


def _dims(x, *vals):
  return torch.tensor(vals, dtype=torch.float32, device=x.device)


def get_gaussian_filter(center, size, lg_var, image_size, filter_size):
  """modellib.py:581-612: center, size, lg_var [B] -> un-normalised filter bank [B, L, F]."""
  return ops.gaussian_filter(center.contiguous().view(-1), size.contiguous().view(-1),
                             lg_var.contiguous().view(-1), int(image_size), int(filter_size))


def extract_patch(x, f_y, f_x, nchannels, normalize=False):
  """modellib.py:615-641: per channel f_y^T . x_d . f_x; x [B,H,W,D], f_y [B,H,FH],
  f_x [B,W,FW] -> [B,FH,FW,D].  (`normalize` is unused in the reference too.)"""
  if x.shape[3] != nchannels:
    x = x[..., :nchannels]
  return ops.extract_patch_dense(x.contiguous(), f_y.contiguous(), f_x.contiguous())


def get_unnormalized_center(ctr_norm, inp_height, inp_width):
  """modellib.py:752-764."""
  return (ctr_norm + 1.0) * (_dims(ctr_norm, inp_height, inp_width) / 2.0)


def get_normalized_center(ctr, inp_height, inp_width):
  """modellib.py:767-779."""
  return ctr / (_dims(ctr, inp_height, inp_width) / 2.0) - 1


def get_unnormalized_size(lg_size, inp_height, inp_width):
  """modellib.py:812-825."""
  return torch.exp(lg_size) * _dims(lg_size, inp_height, inp_width)


def get_normalized_size(size, inp_height, inp_width):
  """modellib.py:828-840."""
  return torch.log(size / _dims(size, inp_height, inp_width))


def get_unnormalized_attn(ctr, lg_size, inp_height, inp_width):
  """modellib.py:843-847."""
  return (get_unnormalized_center(ctr, inp_height, inp_width),
          get_unnormalized_size(lg_size, inp_height, inp_width))


def get_normalized_var(size, filter_height, filter_width):
  """modellib.py:782-793."""
  return torch.log(size) - torch.log(_dims(size, filter_height, filter_width))


def get_normalized_gamma(size, filter_height, filter_width):
  """modellib.py:796-809."""
  return math.log(float(filter_height * filter_width)) - torch.log(size.prod(dim=-1))


def get_box_coord(ctr, size, truncate=True):
  """modellib.py:850-852."""
  return ctr - size / 2.0, ctr + size / 2.0


def get_box_ctr_size(top_left, bot_right):
  """modellib.py:855-856."""
  return (top_left + bot_right) / 2.0, (bot_right - top_left)


def f_segm_match(iou, s_gt):
  """modellib.py:382-415: mask, quantise to 1e-6, Hungarian on iou + 1e-5, re-mask.

  tf.round of TF 0.12 is taken as floor(x + 0.5) (SURVEY.md §8a trap 9)."""
  mask_x = s_gt[:, None, :]
  mask_y = s_gt[:, :, None]
  iou_mask = iou * mask_x * mask_y
  iou_mask = torch.floor(iou_mask * 1e6 + 0.5) / 1e6
  match_eps = hungarian((iou_mask + 1e-5).to(torch.float32))[0]
  return match_eps.to(iou.device) * mask_x * mask_y


# --------------------------------------------------------------------------------------
# Loss / statistics operators (forward).  The pixel-sized contractions and reductions run in
# csrc/ra_loss.hip (one streaming pass, ops.pair_stats / ops.gt_box); what is left here is
# [B,T]- and [B,T,T]-sized bookkeeping with the reference's names and argument lists.
# full_model.get_model(...).run(['loss', 'iou_soft', ...]) uses the fused ra_loss_stats_f32.
# --------------------------------------------------------------------------------------
def _pair(a, b):
  """a [B,N,H,W], b [B,M,H,W] -> (iou [B,N,M], inter [B,N,M], sum_a [B,N,1], sum_b [B,1,M],
  eps*HW).  `inter` is the kernel's own sum(a*b), not recovered from the ratio."""
  st = ops.pair_stats(a, b, want=('iou_soft', 'inter', 'sum_a', 'sum_b'))
  e = 1e-5 * a.shape[2] * a.shape[3]
  sa, sb = st['sum_a'][:, :, None], st['sum_b'][:, None, :]
  return st['iou_soft'], st['inter'], sa, sb, e


def _as4(t):
  """[H,W] / [N,H,W] / [B,N,H,W] -> [B,N,H,W] (the reference accepts all three)."""
  return t.reshape((1,) * (4 - t.dim()) + tuple(t.shape))


def _aligned(m, a, b):
  """The non-pairwise result of an elementwise-broadcast op from its pairwise matrix m [B,N,M]:
  N == M -> the diagonal [B,N]; N == 1 or M == 1 -> the broadcast row / column [B,max(N,M)]
  (the reference multiplies a [B,1,H,W] against b [B,T,H,W], full_model.py:752-754,
  box_model.py:487-490); anything else is a shape error there too."""
  N, M = a.shape[1], b.shape[1]
  if N == M:
    return torch.diagonal(m, dim1=1, dim2=2).reshape(a.shape[:2])
  if N == 1:
    return m[:, 0, :].contiguous()
  if M == 1:
    return m[:, :, 0].contiguous()
  raise RecAttendError('incompatible instance counts %d and %d (broadcast needs equal or 1)' % (N, M))


def f_inter(a, b):
  """modellib.py:107-110 -> [B,N] (broadcast over a singleton instance axis)."""
  a, b = _as4(a), _as4(b)
  return _aligned(_pair(a, b)[1], a, b)


def f_union(a, b, eps=1e-5):
  """modellib.py:113-117 (eps summed over every pixel) -> [B,N]."""
  a, b = _as4(a), _as4(b)
  _, inter, sa, sb, e = _pair(a, b)
  u = sa + sb - inter + eps * a.shape[2] * a.shape[3]
  return _aligned(u, a, b)


def f_iou(a, b, timespan=None, pairwise=False):
  """modellib.py:124-155: pairwise -> [B,N,M]; aligned -> [B,N]."""
  a, b = _as4(a), _as4(b)
  iou = _pair(a, b)[0]
  return iou if pairwise else _aligned(iou, a, b)


def f_dice(a, b, timespan=None, pairwise=False):
  """modellib.py:71-104."""
  a, b = _as4(a), _as4(b)
  _, inter, sa, sb, e = _pair(a, b)
  d = 2.0 * inter / ((sa + e) + (sb + e))
  return d if pairwise else _aligned(d, a, b)


def get_identity_match(num_ex, timespan, s_gt):
  """modellib.py:28-37."""
  eye = torch.eye(timespan, dtype=s_gt.dtype, device=s_gt.device)[None]
  return eye * s_gt[:, None, :] * s_gt[:, :, None]


def f_cum_min(s, d):
  """modellib.py:40-53."""
  return torch.cummin(s, dim=1)[0]


def f_cum_max(s, d):
  """modellib.py:56-68 (cumulative maximum from the END)."""
  return torch.flip(torch.cummax(torch.flip(s, [1]), dim=1)[0], [1])


def f_bce(y_out, y_gt):
  """modellib.py:424-427."""
  eps = 1e-5
  return -y_gt * torch.log(y_out + eps) - (1 - y_gt) * torch.log(1 - y_out + eps)


def f_bce_minmax(y_out_min, y_out_max, y_gt):
  """modellib.py:430-437."""
  eps = 1e-5
  return -y_gt * torch.log(y_out_min + eps) - (1 - y_gt) * torch.log(1 - y_out_max + eps)


def f_coverage(iou):
  """modellib.py:265-274."""
  return iou.max(dim=1)[0]


def f_coverage_weight(y_gt):
  """modellib.py:277-289."""
  s = ops.pair_stats(y_gt, y_gt, want=('sum_b',))['sum_b']
  return s / (s.sum(dim=1, keepdim=True) + (s == 0).to(s.dtype))


def f_weighted_coverage(iou, y_gt):
  """modellib.py:292-302."""
  return (f_coverage(iou) * f_coverage_weight(y_gt)).sum() / float(y_gt.shape[0])


def f_unweighted_coverage(iou, count):
  """modellib.py:305-313."""
  return (f_coverage(iou).sum(dim=1) / count).sum() / float(iou.shape[0])


def f_conf_loss(s_out, match, timespan, use_cum_min=True):
  """modellib.py:316-339."""
  match_sum = match.sum(dim=2)
  if use_cum_min:
    s_bce = f_bce_minmax(f_cum_min(s_out, timespan), f_cum_max(s_out, timespan), match_sum)
  else:
    s_bce = f_bce(s_out, match_sum)
  return s_bce.sum() / float(s_out.shape[0]) / float(s_out.shape[1])


def f_greedy_match(score, matched):
  """modellib.py:365-379."""
  score = score * (1.0 - matched)
  mx = score.max(dim=1, keepdim=True)[0]
  match = (score == mx).to(score.dtype)
  return match / match.sum(dim=1, keepdim=True)


def f_count_acc(s_out, s_gt):
  """modellib.py:482-494."""
  cout = (s_out > 0.5).to(s_out.dtype).sum(dim=1)
  return (cout == s_gt.sum(dim=1)).to(s_out.dtype).sum() / float(s_out.shape[0])


def f_dic(s_out, s_gt, abs=False):
  """modellib.py:497-511."""
  diff = (s_out > 0.5).to(s_out.dtype).sum(dim=1) - s_gt.sum(dim=1)
  if abs:
    diff = diff.abs()
  return diff.sum() / float(s_out.shape[0])


def get_gt_box(y_gt, padding_ratio=0.0, center_shift_ratio=0.0, min_padding=10.0):
  """modellib.py:663-701 -> top_left [B,T,2], bot_right [B,T,2], box [B,T,H,W].  padding_ratio / center_shift_ratio: numbers, or
  tensors broadcastable to [B,T,1] or [B,T,2] — the noisy ground-truth boxes of the training graph (full_model.py:567-577 draws
  them per instance).  Scalars run as one fused launch (ra_gt_box_f32); tensors take the reduction from the same kernel and do
  the [B,T]-sized corner algebra and the box fill (get_filled_box_idx, modellib.py:731-748) as device tensor ops."""
  scalar = isinstance(padding_ratio, (int, float)) and isinstance(center_shift_ratio, (int, float))
  if scalar and center_shift_ratio == 0.0:
    params, box = ops.gt_box(y_gt, float(padding_ratio), float(min_padding))
    return params[:, :, 0:2].contiguous(), params[:, :, 2:4].contiguous(), box
  B, T, H, W = y_gt.shape
  dev = y_gt.device
  as_t = lambda v: (v if isinstance(v, torch.Tensor) else torch.as_tensor(v, dtype=torch.float32)).to(device=dev, dtype=torch.float32)
  pr, cs = as_t(padding_ratio), as_t(center_shift_ratio)
  raw, _ = ops.gt_box(y_gt, 0.0, 0.0, want_box=False)  # the reductions of :679-683 with the empty instances already fixed to 0
  nz = (ops.pair_stats(y_gt, y_gt, want=('sum_b',))['sum_b'] > 0).to(torch.float32)[:, :, None]
  # an empty instance reduces to (H W, H W) / (0, 0) in the reference (idx + (1 - 0) * H W; idx * 0): its box comes out empty
  top_left = raw[:, :, 0:2] * nz + (1.0 - nz) * float(H * W)
  bot_right = raw[:, :, 2:4] * nz
  size = bot_right - top_left
  pad = torch.clamp(pr * size, min=float(min_padding))
  top_left = top_left + cs * size - pad
  bot_right = bot_right + cs * size + pad
  yy = torch.arange(H, dtype=torch.float32, device=dev).view(1, 1, H, 1)
  xx = torch.arange(W, dtype=torch.float32, device=dev).view(1, 1, 1, W)
  tl, br = top_left[:, :, :, None, None], bot_right[:, :, :, None, None]
  box = ((yy >= tl[:, :, 0]) & (xx >= tl[:, :, 1]) & (yy <= br[:, :, 0]) & (xx <= br[:, :, 1])).to(torch.float32)
  top_left = top_left * nz
  bot_right = nz * bot_right + (1.0 - nz) * (2.0 * float(min_padding))
  return top_left.contiguous(), bot_right.contiguous(), box


def get_gt_attn(y_gt, filter_height, filter_width, padding_ratio=0.0, center_shift_ratio=0.0,
                min_padding=10.0):
  """modellib.py:644-660."""
  top_left, bot_right, box = get_gt_box(y_gt, padding_ratio=padding_ratio,
                                        center_shift_ratio=center_shift_ratio,
                                        min_padding=min_padding)
  ctr, size = get_box_ctr_size(top_left, bot_right)
  lg_var = get_normalized_var(size, filter_height, filter_width)
  lg_gamma = get_normalized_gamma(size, filter_height, filter_width)
  return ctr, size, lg_var, lg_gamma, box, top_left, bot_right


def f_iou_box(top_left_a, bot_right_a, top_left_b, bot_right_b):
  """modellib.py:206-238: IoU of axis-aligned boxes from their corners, [B,T,2] each (broadcast over
  T) -> [B,T].  [B,T]-sized bookkeeping; differentiable."""
  y1a, x1a, y2a, x2a = top_left_a[..., 0], top_left_a[..., 1], bot_right_a[..., 0], bot_right_a[..., 1]
  y1b, x1b, y2b, x2b = top_left_b[..., 0], top_left_b[..., 1], bot_right_b[..., 0], bot_right_b[..., 1]
  x1, y1 = torch.maximum(x1a, x1b), torch.maximum(y1a, y1b)
  x2, y2 = torch.minimum(x2a, x2b), torch.minimum(y2a, y2b)
  flag = (x1 < x2).to(x1.dtype) * (y1 < y2).to(x1.dtype)
  inter = flag * (x2 - x1) * (y2 - y1)
  return inter / ((x2a - x1a) * (y2a - y1a) + (x2b - x1b) * (y2b - y1b) - inter)


def f_squared_err(y_out, y_gt):
  """modellib.py:525-530."""
  err = y_out - y_gt
  return 0.5 * err * err


def f_huber(y_out, y_gt, threshold=1.0):
  """modellib.py:514-522 (note the reference's indicator is `err <= 1`, not |err| <= threshold)."""
  err = y_out - y_gt
  ind = (err <= 1).to(err.dtype)
  return 0.5 * err * err * ind + (err.abs() - (threshold - 0.5 * threshold ** 2)) * (1 - ind)


def f_match_loss(y_out, y_gt, match, timespan, loss_fn, model=None):
  """modellib.py:440-478: sum_ij match[b,i,j] * sum_d loss_fn(y_out[b,i,:], y_gt[b,j,:]), divided by
  the match count, the batch size and the feature size.  [B,N,D] inputs (the attention parameters,
  full_model.py:891-892,952-964); the [B,N,H,W] use never executes in the reference (its only call
  sites are `box_loss_fn = ...f_bce` — an assignment to the wrong name, :971 — and the undefined
  f_match_bce, :1016)."""
  if y_out.dim() != 3:
    raise NotImplementedError('f_match_loss on [B,N,H,W] inputs is dead code in the reference (full_model.py:971,1016)')
  B, N, D = y_out.shape
  pair = loss_fn(y_out[:, :, None, :], y_gt[:, None, :, :]).sum(dim=3)      # [B,N,N]
  count = torch.clamp(match.sum(dim=(1, 2)), min=1.0)
  return ((pair * match).sum(dim=(1, 2)) / count).sum() / float(B) / float(D)


def _not_built(name):
  def fn(*a, **k):
    raise NotImplementedError('modellib.%s belongs to the training step (SURVEY.md §8f rank 2) '
                              'and is not built yet' % name)
  fn.__name__ = name
  return fn


for _n in ('f_sem_loss',):
  globals()[_n] = _not_built(_n)
