"""utils/postprocess.py of the reference on the MI355X: same function names and argument
meaning, float32 CUDA tensors [B,T,H,W] instead of lists of numpy arrays.  `postprocess`
(below) is the fused form full_model_eval.py:112-124 reduces to when the cv2 steps are skipped."""
import torch

import ra_ops as ops


def apply_threshold(y_out, thresh):
  """postprocess.py:5-12."""
  return (y_out > thresh).to(torch.float32)


def apply_confidence(y_out, s_out):
  """postprocess.py:15-29 -> (y_out * s_out, s_out > 0.5)."""
  return y_out * s_out[:, :, None, None], (s_out > 0.5).to(torch.float32)


def apply_one_label(y_out):
  """postprocess.py:32-52: every pixel keeps only its arg-max instance (first maximum)."""
  ones = torch.ones(y_out.shape[:2], dtype=torch.float32, device=y_out.device)
  keep, _, _ = ops.postprocess(y_out, ones, float('-inf'))
  return keep * y_out


def mask_foreground(y_out, fg):
  """postprocess.py:139-147."""
  return y_out * fg[:, None]


def remove_tiny(y_out, conf, threshold=200):
  """postprocess.py:109-136 (returns new tensors; y_out binary or soft)."""
  if threshold == 0:
    return y_out, conf
  sizes = ops.pair_stats(y_out, y_out[:, :1], want=('sum_a',))['sum_a']
  y_out, conf = y_out.clone(), conf.clone().to(torch.float32)
  ops.remove_tiny(y_out, sizes, conf, float(threshold))
  return y_out, conf


def morph(y_out):
  """postprocess.py:55-72: cv2.dilate of every instance plane with a 5 x 5 box (out-of-image pixels ignored)."""
  return ops.dilate(y_out, 2)


def upsample(y_out, y_gt):
  """postprocess.py:75-106: every plane of y_out [B,T,H',W'] resized to y_gt's [H,W] (cv2.resize, INTER_LINEAR) and passed
  through cv2.bilateralFilter(b, 5, 10, 10).  The resize is cv2's float32 arithmetic; the bilateral filter evaluates the
  exponentials cv2 reads from an interpolated table (cv2 is not part of this stack, so the two have not been compared: the
  step is optional in the evaluator and off by default)."""
  H, W = (y_gt.shape[-2], y_gt.shape[-1]) if hasattr(y_gt, 'shape') else (int(y_gt[0]), int(y_gt[1]))
  return ops.bilateral5(ops.resize_linear(y_out, H, W), 10.0, 10.0)


def postprocess(y_out, s_out, thresh, fg=None, remove_tiny_threshold=0):
  """full_model_eval.py:112-124 without the cv2 steps, one pass: apply_confidence ->
  apply_one_label -> apply_threshold [-> mask_foreground -> remove_tiny].
  Returns (y_out_thresh [B,T,H,W] binary, s_out_hard [B,T], union [B,H,W])."""
  if s_out.dim() == 3:
    s_out = s_out[:, :, 0].contiguous()  # multi-class: full_model_eval.py:108-110
  y_bin, s_hard, uni = ops.postprocess(y_out, s_out, float(thresh), fg=fg, want_union=True)
  if fg is not None and remove_tiny_threshold:
    sizes = ops.pair_stats(y_bin, y_bin[:, :1], want=('sum_a',))['sum_a']
    ops.remove_tiny(y_bin, sizes, s_hard, float(remove_tiny_threshold))
    uni = None
  return y_bin, s_hard, uni
