"""The optimisation step of the reference's training graph (full_model.py:1039-1057) and its
data-parallel form: ONE flat float32 gradient bucket, ONE all-reduce (RCCL on
MI355X, gloo in the CPU tests) per step, then clip + Adam in one HIP pass.

  bucket = GradBucket(model)            # parameters become views of one flat buffer
  ... backward fills bucket.grad (views: bucket.grad_of[name]) ...
  bucket.allreduce()                    # sum over ranks (no-op for world 1)
  bucket.step()                         # g/world -> +wd*w -> clip(+-1) -> Adam(eps 1e-7), global_step += 1

Sum-then-scale-then-clip reproduces the single-process update on the concatenated batch when every
rank's loss is divided by the GLOBAL example count (num_ex_f, full_model.py:916), which is how
TrainStep scales it.  BatchNorm batch moments (nnlib.py:98) are all-reduced separately
(allreduce_moments) so that the normalisation equals the single-process one too.
"""
import math

import numpy as np
import torch

import ra_native as rn
from ra_native import check, ptr

BETA1, BETA2, ADAM_EPS, CLIP = 0.9, 0.999, 1e-7, 1.0  # tf.train.AdamOptimizer defaults; full_model.py:1046,1053


def learn_rate(opt, global_step):
  """tf.train.exponential_decay(base, step, steps_per_decay, decay, staircase=True)
  (full_model.py:1039-1045)."""
  k = int(global_step) // int(opt['steps_per_learn_rate_decay'])
  return float(opt['base_learn_rate']) * float(opt['learn_rate_decay']) ** k


def knob_prob(opt, global_step, offset):
  """tf.train.exponential_decay(knob_base, max(0, step - offset), steps_per_knob_decay, knob_decay)
  without staircase (full_model.py:601-623)."""
  s = max(0.0, float(global_step) - float(offset))
  return float(opt['knob_base']) * float(opt['knob_decay']) ** (s / float(opt['steps_per_knob_decay']))


def is_decayed(name):
  """nnlib registers wd * l2_loss only for `w` tensors (nnlib.py:59-61,206-207,333,471,611-618):
  filters and matrices, not biases and not BatchNorm parameters."""
  tail = name.split('_')
  return ('w' in tail) or any(t.startswith('w') and len(t) == 3 and t[1] in 'xh' for t in tail[-1:])


def trainable(name):
  """BN EMA shadows are not trainable (nnlib.py:121-127: updated by assignment, not by Adam)."""
  return not (name.endswith('_ema_mean') or name.endswith('_ema_var'))


class GradBucket(object):
  """Flat parameter / gradient / Adam-moment buffers; model[name] tensors become views."""

  def __init__(self, model, opt=None, device=None, names=None):
    self.model = model
    self.opt = dict(opt if opt is not None else getattr(model, 'opt', {}))
    names = names if names is not None else [k for k in model.weight_keys() if trainable(k)]
    self.names = sorted(names)
    dev = device if device is not None else model[self.names[0]].device
    self.offsets, off = {}, 0
    for k in self.names:
      n = model[k].numel()
      self.offsets[k] = (off, n, tuple(model[k].shape))
      off += (n + 3) & ~3  # 16-byte aligned views
    self.n = off
    self.param = torch.zeros(off, dtype=torch.float32, device=dev)
    self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
    self.m = torch.zeros(off, dtype=torch.float32, device=dev)
    self.v = torch.zeros(off, dtype=torch.float32, device=dev)
    wd = float(self.opt.get('weight_decay', 0.0) or 0.0)
    self.wd = torch.zeros(off, dtype=torch.float32, device=dev)
    self.grad_of = {}
    for k in self.names:
      o, n, shp = self.offsets[k]
      self.param[o:o + n].copy_(model[k].reshape(-1))
      model[k] = self.param[o:o + n].view(shp)  # the model now reads the bucket's memory
      self.grad_of[k] = self.grad[o:o + n].view(shp)
      if wd and is_decayed(k):
        self.wd[o:o + n] = wd
    self.global_step = int(model.get('global_step', 0) or 0)

  def zero_grad(self):
    self.grad.zero_()

  def allreduce(self):
    """One collective per step over the whole bucket (2.3 MiB at the CVPPP arch: latency-bound on
    xGMI).  Returns the world size the sum ran over."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
      return 1
    dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)
    return dist.get_world_size()

  def step(self, world=1, lr=None):
    """clip(grad / world + wd * w, +-1) -> Adam; increments global_step (full_model.py:1048-1056)."""
    t = self.global_step + 1
    lr = learn_rate(self.opt, self.global_step) if lr is None else lr
    lr_t = lr * math.sqrt(1.0 - BETA2 ** t) / (1.0 - BETA1 ** t)
    if not self.param.is_cuda:
      raise rn.RecAttendError('GradBucket.step runs the HIP optimizer kernel; no CPU fallback')
    import ctypes as C
    check(rn.lib().ra_adam_step_f32(ptr(self.param), ptr(self.grad), ptr(self.m), ptr(self.v), ptr(self.wd),
                                    self.n, C.c_float(lr_t), C.c_float(BETA1), C.c_float(BETA2),
                                    C.c_float(ADAM_EPS), C.c_float(CLIP), C.c_float(1.0 / world),
                                    rn.stream_ptr()), 'ra_adam_step_f32')
    self.global_step = t
    self.model['global_step'] = float(t)
    return lr


def allreduce_moments(sums):
  """BatchNorm batch moments over the GLOBAL batch (nnlib.py:98 normalises over the whole batch):
  `sums` [.., 2C+1] = per-channel (sum x, sum x^2) and the element count, summed over the ranks."""
  import torch.distributed as dist
  if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
  return sums
