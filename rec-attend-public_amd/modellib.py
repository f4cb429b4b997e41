"""`modellib` surface of the reference on gfx950 kernels (cited as modellib.py:line).

Built: the Gaussian-attention operators of the decode loop (get_gaussian_filter,
extract_patch and the (un)normalisers), the Hungarian op binding and f_segm_match's
conditioning around it.  The loss / IoU family (modellib.py:28-531) belongs to the training
step, SURVEY.md §8(f) rank 2, and is not built yet.
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


def _not_built(name):
  def fn(*a, **k):
    raise NotImplementedError('modellib.%s belongs to the training step (SURVEY.md §8f rank 2) '
                              'and is not built yet' % name)
  fn.__name__ = name
  return fn


for _n in ('f_iou', 'f_dice', 'f_inter', 'f_union', 'f_iou_box', 'f_weighted_coverage',
           'f_unweighted_coverage', 'f_conf_loss', 'f_greedy_match', 'f_match_loss', 'f_bce',
           'f_bce_minmax', 'f_cum_min', 'f_cum_max', 'f_count_acc', 'f_dic', 'get_gt_attn',
           'get_gt_box', 'get_identity_match'):
  globals()[_n] = _not_built(_n)
