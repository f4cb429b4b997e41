"""The optimisation step of the reference's training graph (full_model.py:1039-1057) and its
data-parallel form: ONE flat float32 gradient bucket, ONE all-reduce (RCCL on
MI355X, gloo in the CPU tests) per step, then clip + Adam in one HIP pass.

  bucket = GradBucket(model)            # parameters become views of one flat buffer
  ... backward fills bucket.grad (views: bucket.grad_of[name]) ...
  bucket.allreduce()                    # sum over ranks (no-op for world 1)
  bucket.step()                         # g/world -> +wd*w -> clip(+-1) -> Adam(eps 1e-7), global_step += 1

Sum-then-scale-then-clip reproduces the single-process update on the concatenated batch: every rank's
loss divides by ITS OWN example count (num_ex_f, full_model.py:916 on the shard), the bucket is summed
over the ranks and the optimizer kernel scales it by 1 / world (equal shards: batch_size % world == 0 is
required by the trainers).  BatchNorm batch moments (nnlib.py:98) are taken over the rank's shard by
default; with model_opt['sync_bn'] every BN call gathers the ranks' (count, mean, M2) and its backward
all-reduces the two per-channel sums, so that normalisation and gradient equal the single-process ones
(sync_moments below).  Initial weights, Adam moments, EMA shadows and global_step are
broadcast from rank 0 when a trainer is built (broadcast_state), so ranks start from ONE model whatever
their random seeds were.
"""
import contextlib
import math
import os

import numpy as np
import torch

import ra_native as rn
from ra_native import check, ptr

BETA1, BETA2, ADAM_EPS, CLIP = 0.9, 0.999, 1e-7, 1.0  # tf.train.AdamOptimizer defaults; full_model.py:1046,1053


def learn_rate(opt, global_step):
  """tf.train.exponential_decay(base, step, steps_per_decay, decay, staircase=True)
  (full_model.py:1039-1045)."""
  # the reference CLI's defaults where a hand-built model_opt omits them (full_model_train.py:481-484)
  k = int(global_step) // int(opt.get('steps_per_learn_rate_decay', 5000))
  return float(opt.get('base_learn_rate', 0.001)) * float(opt.get('learn_rate_decay', 0.96)) ** k


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
    # four more floats behind the gradients: [off] = "a status word of this step is non-zero on this rank" — it rides on the
    # bucket's all-reduce, so after the sum every rank knows whether ANY rank's step failed, and the guarded optimizer
    # kernel skips the update on all of them together (no extra collective)
    self.grad_full = torch.zeros(off + 4, dtype=torch.float32, device=dev)
    self.grad = self.grad_full[:off]
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
    self.grad_full.zero_()

  def allreduce(self):
    """One collective per step over the whole bucket (2.3 MiB at the CVPPP arch: latency-bound on
    xGMI).  Returns the world size the sum ran over."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
      return 1
    dist.all_reduce(self.grad_full, op=dist.ReduceOp.SUM)
    return dist.get_world_size()

  def broadcast(self, src=0):
    """Every rank takes rank `src`'s parameters, Adam moments and global_step: get_model() draws the initial
    weights from each process's own generator, and a data-parallel step sums gradients — of ONE model only
    if the ranks hold the same weights.  No-op for world 1."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
      return 1
    for t in (self.param, self.m, self.v):
      dist.broadcast(t, src=src)
    gs = torch.tensor([float(self.global_step)], dtype=torch.float64, device=self.param.device if dist.get_backend() == 'nccl' else 'cpu')
    dist.broadcast(gs, src=src)
    self.global_step = int(gs.item())
    self.model['global_step'] = float(self.global_step)
    return dist.get_world_size()

  def state_dict(self):
    """What utils/saver.py:24-31 saves beside the weights (tf.all_variables(): the Adam slots and global_step),
    by parameter name, as NumPy arrays."""
    out = {'global_step': np.asarray(self.global_step, dtype=np.int64)}
    for k in self.names:
      o, n, shp = self.offsets[k]
      out['adam_m/' + k] = self.m[o:o + n].view(shp).detach().cpu().numpy()
      out['adam_v/' + k] = self.v[o:o + n].view(shp).detach().cpu().numpy()
    return out

  def load_state_dict(self, state, strict=True):
    for k in self.names:
      o, n, shp = self.offsets[k]
      for slot, buf in (('adam_m/', self.m), ('adam_v/', self.v)):
        if slot + k in state:
          v = torch.as_tensor(np.asarray(state[slot + k], dtype=np.float32))
          if tuple(v.shape) != tuple(shp):
            raise rn.RecAttendError('optimizer state %s%s: shape %r != %r' % (slot, k, tuple(v.shape), tuple(shp)))
          buf[o:o + n].copy_(v.reshape(-1))
        elif strict:
          raise rn.RecAttendError('optimizer state lacks %s%s' % (slot, k))
    if 'global_step' in state:
      self.global_step = int(np.asarray(state['global_step']))
      self.model['global_step'] = float(self.global_step)
    elif strict:
      raise rn.RecAttendError('optimizer state lacks global_step')

  def step(self, world=1, lr=None, status=None, n_solver=0):
    """clip(grad / world + wd * w, +-1) -> Adam; increments global_step (full_model.py:1048-1056).  status: an int32
    device tensor of this step's status words, the first n_solver of them Hungarian-solver codes (failed when
    negative), the rest failed when non-zero — the update is applied only if none says failed (ra_adam_step_guarded_f32), so a step whose matching failed never reaches the weights even though the host looks at
    the words one step late."""
    t = self.global_step + 1
    lr = learn_rate(self.opt, self.global_step) if lr is None else lr
    lr_t = lr * math.sqrt(1.0 - BETA2 ** t) / (1.0 - BETA1 ** t)
    if not self.param.is_cuda:
      raise rn.RecAttendError('GradBucket.step runs the HIP optimizer kernel; no CPU fallback')
    import ctypes as C
    if status is not None and status.numel():
      assert status.dtype == torch.int32 and status.is_contiguous() and status.device == self.param.device
      check(rn.lib().ra_adam_step_guarded_f32(ptr(self.param), ptr(self.grad), ptr(self.m), ptr(self.v), ptr(self.wd),
                                              self.n, C.c_float(lr_t), C.c_float(BETA1), C.c_float(BETA2),
                                              C.c_float(ADAM_EPS), C.c_float(CLIP), C.c_float(1.0 / world),
                                              ptr(status), int(n_solver), int(status.numel()) - int(n_solver), rn.stream_ptr()),
            'ra_adam_step_guarded_f32')
    else:
      check(rn.lib().ra_adam_step_f32(ptr(self.param), ptr(self.grad), ptr(self.m), ptr(self.v), ptr(self.wd),
                                      self.n, C.c_float(lr_t), C.c_float(BETA1), C.c_float(BETA2),
                                      C.c_float(ADAM_EPS), C.c_float(CLIP), C.c_float(1.0 / world),
                                      rn.stream_ptr()), 'ra_adam_step_f32')
    self.global_step = t
    self.model['global_step'] = float(t)
    eng = getattr(self.model, 'engine', None)
    if eng is not None:  # the kernel wrote the weights behind torch's version counters: repack on next decode
      eng._stamp = None
    return lr


def combine_moments(count, mean, var):
  """Whole-batch moments from per-shard ones: count [R], mean / var [R,C] (biased variance, as tf.nn.moments) ->
  (total count, mean [C], var [C]) by Chan's pairwise update (no E[x^2] - E[x]^2 cancellation)."""
  count = count.to(mean.dtype)
  n = count.sum()
  w = (count / n)[:, None]
  m = (w * mean).sum(dim=0)
  v = (w * (var + (mean - m[None, :]) ** 2)).sum(dim=0)
  return n, m, v


def sync_moments(mean, var, n_local):
  """BatchNorm batch moments over the GLOBAL batch (nnlib.py:98 normalises over the whole batch): every rank
  contributes (count, mean, var) of its shard in ONE all_gather of 2C+1 floats; mean / var are overwritten with
  the whole-batch moments.  Returns the global element count (n_local for world 1)."""
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return float(n_local)
  C = mean.numel()
  mine = torch.empty(2 * C + 1, dtype=torch.float32, device=mean.device)
  mine[:C], mine[C:2 * C], mine[2 * C] = mean, var, float(n_local)
  parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
  dist.all_gather(parts, mine)
  allp = torch.stack(parts)
  n, m, v = combine_moments(allp[:, 2 * C], allp[:, :C], allp[:, C:2 * C])
  mean.copy_(m)
  var.copy_(v)
  return float(n_local) * dist.get_world_size()  # equal shards (the trainers require batch_size % world == 0): no host sync


def allreduce_sums(t):
  """Sum of a small tensor over the ranks (the 2C per-channel sums of a synchronised BatchNorm backward)."""
  import torch.distributed as dist
  if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
  return t


# =====================================================================================================
# The training graph (full_model.py:638-1057, phase_train = True) as a torch.autograd tape over
# HIP kernels.  torch is the tape and the memory; what runs on the GPU:
#   * every conv layer of the three CNNs — forward (MFMA conv, BN batch moments, normalise + ReLU +
#     pool) and backward (BN / pool / ReLU adjoint, backward-data on the MFMA conv kernel with the
#     flipped packing, backward-weight on MFMA) — ConvBNActPool, csrc/ra_train.hip + ra_conv.hip;
#   * the pairwise soft IoU of the two losses and its adjoint — PairIoU, csrc/ra_loss.hip;
#   * the ground-truth boxes and both Hungarian matchings (device solver), clip + Adam;
#   * the attention resample — box, read, write — forward on the decode loop's banded kernels and backward
#     straight to the window parameters' gradients (AttnExtract / AttnPaste, csrc/ra_attn_train.hip): no dense
#     [L,F] filter banks, no GEMMs;
#   * the controller's small dense algebra (LSTM / MLP GEMMs on [B, <= 1024] tensors) as library GEMMs +
#     elementwise ops under autograd.
# =====================================================================================================
import ctypes as _C

import ra_ops as ops

BN_EPS = ops.BN_EPS
EMA_DECAY = 0.9  # 1 - 0.1 * phase_train (nnlib.py:103-104)


_POISON = os.environ.get('RA_POISON')  # a debugging aid: every scratch tensor starts as this value (e.g. "nan") instead of stale memory


def _f(*shape, device):
  if _POISON:
    return torch.full(shape, float(_POISON), dtype=torch.float32, device=device)
  return torch.empty(shape, dtype=torch.float32, device=device)


_SIDE = {}


def _side_stream(device, which=0):
  k = (str(device), which)
  if k not in _SIDE:
    _SIDE[k] = torch.cuda.Stream(device=device)
  return _SIDE[k]


class _LatestVersionPack(dict):
  """The pack cache of callers OUTSIDE a TrainStep (the train-mode nnlib closures, tests, one-off layer calls).  Keys carry the
  source tensor's data pointer AND version — (ptr, version, geometry...), ('shift', ptr, version, cp), ('split', ptr, version,
  transposed) — so an in-place optimizer step makes every entry of that weight stale; a plain dict kept them all and an
  ordinary training loop over the closures grew by one set of packed filters per layer, direction and step (~1 MB
  per step at the CVPPP arch).  Here a new version of the same (tensor, geometry) REPLACES the old entry, and the whole cache
  is capped (oldest first) for callers that keep creating tensors."""
  CAP = 2048

  def __init__(self):
    super().__init__()
    self._latest = {}

  @staticmethod
  def _identity(key):
    return (key[0], key[1]) + tuple(key[3:]) if isinstance(key[0], str) else (key[0],) + tuple(key[2:])

  def __setitem__(self, key, value):
    ident = self._identity(key)
    old = self._latest.get(ident)
    if old is not None and old != key:
      super().pop(old, None)
    self._latest[ident] = key
    super().__setitem__(key, value)
    while len(self) > self.CAP:
      k0 = next(iter(self))
      super().pop(k0)
      self._latest.pop(self._identity(k0), None)

  def clear(self):
    super().clear()
    self._latest.clear()


_PACK = _LatestVersionPack()
# A TrainStep owns its own cache
# (TrainStep._pack, handed to the layer functions through meta['cache']): per step, (weight storage, geometry) -> packed
# filter — the filters are shared by the T timesteps.  forward_loss() clears the trainer's dict (the optimizer writes the
# weights through raw pointers); an entry keeps its source tensor alive and carries the tensor's version, so a caller
# cannot be handed the pack of a freed tensor whose address was reused, nor one that predates an in-place update.


class _PackCache(dict):
  """A TrainStep's per-step cache of packed filters (cleared by every forward_loss), plus the step's PRE-PACKED filters:
  the device packer only moves values, so packing an array of flat-bucket POSITIONS once per (layer, geometry) gives an
  index map, and every later step packs all of its filters and pads all of its biases with ONE gather launch over the
  parameter bucket (ra_gather_f32) instead of one pack launch (+ a fill and a copy for a padded bias) per layer and
  direction.  A geometry seen for the first time takes the per-layer launch and joins the map for the next step."""

  def __init__(self, trainer):
    super().__init__()
    self.tr = trainer
    self.entries, self.pending = {}, []   # key -> (start, n) in self.wp; [(key, index map)] waiting for the next refresh
    self.imap = self.wp = None
    self.filled = -1                        # the pack epoch self.wp holds

  def _where(self, t):
    """Offset (in floats) of a contiguous float32 parameter inside the bucket, or None."""
    par = self.tr.bucket.param
    off = t.data_ptr() - par.data_ptr()
    if t.dtype != torch.float32 or not t.is_contiguous() or off < 0 or off % 4 or off // 4 + t.numel() > par.numel():
      return None
    return off // 4

  def _positions(self, t, off):
    assert self.tr.bucket.param.numel() < (1 << 24)  # positions (+1; 0 = "no source") are exact in float32
    return (torch.arange(t.numel(), device=t.device, dtype=torch.float32) + float(off + 1)).reshape(t.shape)

  def lookup(self, key, t, make_map):
    """key: the packing's geometry; t: the parameter; make_map(positions) -> packed positions.  The packed values of
    this step, or None (first sight of the geometry, or a tensor outside the bucket: the caller packs as before)."""
    off = self._where(t)
    if off is None or not getattr(self.tr, 'prepack', True):
      return None
    key = (off, t.numel()) + key
    hit = self.entries.get(key)
    if hit is not None:
      return self.wp[hit[0]:hit[0] + hit[1]] if self.filled == self.tr._pack_epoch else None
    if all(k != key for k, _ in self.pending) and not torch.cuda.is_current_stream_capturing():
      with torch.no_grad():
        self.pending.append((key, (make_map(self._positions(t, off)).reshape(-1) - 1.0).to(torch.int32)))
    return None

  def merge(self):
    """Take in the geometries met since the last merge.  Never inside a stream capture: the index map is built by eager
    launches from tensors that are freed here (a captured copy of those launches would re-read freed memory at every
    replay); TrainStep._graphed merges before it captures."""
    if self.pending and not torch.cuda.is_current_stream_capturing():
      self.tr._drop_captured_steps()  # the buffers below move
      n0 = 0 if self.imap is None else self.imap.numel()
      maps = ([self.imap] if n0 else [])
      for key, m in self.pending:
        pad = (-m.numel()) % 4       # 16-byte aligned views
        self.entries[key] = (n0, m.numel())
        maps.append(torch.nn.functional.pad(m, (0, pad), value=-1))
        n0 += m.numel() + pad
      self.imap = torch.cat(maps)
      self.wp = torch.empty(n0, dtype=torch.float32, device=self.imap.device)
      self.pending = []

  def refresh(self):
    """Once per optimisation step, after the epoch moved: this step's packed filters and padded biases, one gather."""
    self.merge()
    if self.imap is not None and getattr(self.tr, 'prepack', True):
      check(rn.lib().ra_gather_f32(ptr(self.tr.bucket.param), ptr(self.imap), self.imap.numel(), ptr(self.wp), rn.stream_ptr()),
            'ra_gather_f32')
      self.filled = self.tr._pack_epoch


def _pack_dev(w, cin_w, cout, cin, cmap_t, transposed, cache=None):
  cache = _PACK if cache is None else cache
  key = (w.data_ptr(), w._version, int(cin_w), int(cout), int(cin), 0 if cmap_t is None else cmap_t.data_ptr(), bool(transposed))
  hit = cache.get(key)
  if hit is not None:
    return hit[1]
  if isinstance(cache, _PackCache):
    pre = cache.lookup(('w',) + key[2:], w, lambda pos: _pack_dev_now(pos, cin_w, cout, cin, cmap_t, transposed))
    if pre is not None:
      cache[key] = (w, pre)
      return pre
  out = _pack_dev_now(w, cin_w, cout, cin, cmap_t, transposed)
  cache[key] = (w, out)
  return out


def _pack_dev_now(w, cin_w, cout, cin, cmap_t, transposed):
  n = rn.lib().ra_conv_packed_floats(cin, cout)
  if n == 0:
    raise rn.RecAttendError('unsupported conv shape Cin=%d Cout=%d' % (cin, cout))
  out = _f(n, device=w.device)
  check(rn.lib().ra_conv_pack_weights_dev(ptr(w), int(cin_w), int(cout), int(cin), ptr(cmap_t),
                                          rn.RA_CONV_TRANSPOSED if transposed else 0, ptr(out), rn.stream_ptr()),
        'ra_conv_pack_weights_dev')
  return out


_CONST = {}


def _const(kind, key, device, make):
  """Small constant device tensors (unit scales, zero shifts, channel maps) are built once: a training
  step runs ~1300 layer calls and each host->device upload costs more than the kernel it feeds."""
  k = (kind, key, str(device))
  t = _CONST.get(k)
  if t is None:
    t = _CONST[k] = make()
  return t


def _pad_map(n_real, n_kernel, device):
  """chan_map for a kernel input of n_kernel channels whose first n_real are real."""
  if n_real == n_kernel:
    return None
  return _const('padmap', (n_real, n_kernel), device, lambda: torch.tensor(
      list(range(n_real)) + [-1] * (n_kernel - n_real), dtype=torch.int32, device=device))


def _pad_channels(t, mult=4):
  c = t.shape[-1]
  cp = -(-c // mult) * mult
  if cp == c:
    return t.contiguous()
  out = torch.zeros(t.shape[:-1] + (cp,), dtype=t.dtype, device=t.device)
  out[..., :c] = t
  return out


EPILOGUE_MOMENTS = {'on': os.environ.get('RA_EPI_MOMENTS', '1') != '0'}  # tuning aid: 0 = the two-pass moments kernels


class _DeferredWgrads(dict):
  """(scope, layer) -> the (x, du) pairs of the layer's calls in the running backward pass.  A filter shared by the
  step's T timesteps (every nnlib.cnn / dcnn layer of full_model, full_model.py:455-535) sums T filter gradients; as
  the end-of-backward callback this object forms each layer's sum in ONE pass over the images of all its calls
  (ra_conv3x3_wgrad_multi_acc_f32 through two pointer tables): 4 launches per layer and step instead of 2 T, and the
  patch-sized layers' fixed costs (launch, cross-wave reduction, partial sums) are paid once, not T times."""
  armed = False

  def reset(self):
    """A new forward pass: nothing of an interrupted backward pass may be added to."""
    self.armed = False
    for slot in self.values():
      slot['calls'] = []

  def __call__(self):
    self.armed = False
    for slot in self.values():
      calls, slot['calls'] = slot['calls'], []
      Cx, cout, B, Hs, Ws, H, W, ups, _, cin_w, tr, bf = slot['shape']
      for k in range(0, len(calls), 64):
        part = calls[k:k + 64]
        n = len(part)
        nws = rn.lib().ra_conv3x3_wgrad_workspace_floats(Cx, cout, n * B, H, W)
        if slot['ws'] is None or slot['ws'].numel() < nws:
          slot['ws'] = _f(nws, device=slot['tab'].device)
        tab = slot['tab']
        for off, col in ((0, 0), (64, 1)):
          host = (_C.c_void_p * n)(*[c[col].data_ptr() for c in part])
          check(rn.lib().ra_ptr_table(host, n, tab.data_ptr() + 8 * off, rn.stream_ptr()), 'ra_ptr_table')
        check(rn.lib().ra_conv3x3_wgrad_multi_acc_f32(tab.data_ptr(), tab.data_ptr() + 8 * 64, n, Cx, B, Hs, Ws, ups, cout,
                                                      ptr(slot['ws']), slot['ws'].numel(), ptr(slot['cmap_t']), cin_w, tr,
                                                      ptr(slot['gw']), ptr(slot['gb']), int(bf), rn.stream_ptr()),
              'ra_conv3x3_wgrad_multi_acc_f32')


class ConvBNActPool(torch.autograd.Function):
  """One layer of nnlib.cnn (nnlib.py:229-253) or nnlib.dcnn (nnlib.py:362-400) in training mode.

  x [B,Hs,Ws,Cx] (Cx % 4 == 0), w in the reference layout ([3,3,Cin,Cout]; transposed:
  [3,3,Cout,Cin]), b [Cout], gamma / beta [Cout] or None.  Returns (y, batch mean, batch var)."""

  @staticmethod
  def forward(ctx, x, w, b, gamma, beta, meta):
    ctx.set_materialize_grads(False)  # nothing differentiates mean / var: no zero tensors for their gradients
    dev = x.device
    x = x.contiguous()
    B, Hs, Ws, Cx = x.shape
    tr, stride, pool, relu = meta['transposed'], meta['stride'], meta['pool'], meta['relu']
    cout, cin_w = (w.shape[2], w.shape[3]) if tr else (w.shape[3], w.shape[2])
    cmap = meta.get('chan_map')
    cmap_t = _const('cmap', tuple(cmap), dev, lambda: torch.tensor(cmap, dtype=torch.int32, device=dev)) \
        if cmap is not None else _pad_map(cin_w, Cx, dev)
    cp = ops.cout_padded(cout)
    cache = meta.get('cache')
    cache = _PACK if cache is None else cache
    wp = _pack_dev(w.contiguous(), cin_w, cout, Cx, cmap_t, tr, cache)
    scale = _const('ones', cp, dev, lambda: torch.ones(cp, dtype=torch.float32, device=dev))
    shift = b.detach()
    if cp != cout:  # padded once per step (the layer's bias is shared by all timesteps), not once per use
      key = ('shift', b.data_ptr(), b._version, cp)
      hit = cache.get(key)
      if hit is None:
        pre = cache.lookup(('b', cp), b.detach(), lambda pos: torch.nn.functional.pad(pos, (0, cp - cout))) \
            if isinstance(cache, _PackCache) else None
        hit = cache[key] = (b, pre if pre is not None else torch.nn.functional.pad(b.detach(), (0, cp - cout)))
      shift = hit[1]
    use_bn = gamma is not None
    bf = bool(meta.get('bf16'))
    mean = var = None
    if use_bn:
      mean, var = meta['stat_out'] if meta.get('stat_out') is not None else (_f(cout, device=dev), _f(cout, device=dev))
    alloc = meta.get('alloc')  # the batched-backward step keeps u / y of a layer's T calls in one [T, ...] slab
    place = (lambda kind, *shape: alloc(kind, shape)) if alloc is not None else (lambda kind, *shape: _f(*shape, device=dev))
    up = 2 if stride == 2 else 1
    if meta.get('bf16_store') and alloc is not None and use_bn and not meta.get('sync_bn'):
      # bf16 mode, stacked step: u and y live in HBM as bf16 where the float4 BatchNorm kernels take the channel count (else
      # float32; y also float32 where a float32 kernel reads it next: meta['y_f32']); the statistics come from the conv's
      # float32 accumulators, before the rounding.  Forward only: the stacked graph's nodes hold the backward.
      assert bf and not torch.is_grad_enabled()
      st_ok = cout % 4 == 0 and (cout // 4) & (cout // 4 - 1) == 0 and cout // 4 <= 64 and EPILOGUE_MOMENTS['on']
      ud = torch.bfloat16 if st_ok else torch.float32
      yd = torch.bfloat16 if (st_ok and not meta.get('y_f32')) else torch.float32
      u = alloc('u', (B, Hs * up, Ws * up, cout), ud)
      H, W = u.shape[1], u.shape[2]
      xin = 1 if x.dtype == torch.bfloat16 else 0
      if cout % 4 == 0 and EPILOGUE_MOMENTS['on']:
        part = _f(rn.lib().ra_conv3x3_moments_part_floats(cout), device=dev)
        nparts = _C.c_int(0)
        check(rn.lib().ra_conv3x3_bf16_f32(ptr(x), Cx, None, 0, B, Hs, Ws, int(stride == 2), ptr(wp), ptr(scale), ptr(shift), cout, 0, 1,
                                           ptr(u), ptr(part), part.numel(), _C.byref(nparts), xin | (2 if st_ok else 0),
                                           rn.stream_ptr()), 'ra_conv3x3_bf16_f32')
        check(rn.lib().ra_bn_moments_from_partials_f32(ptr(part), nparts.value, cout, ptr(mean), ptr(var), rn.stream_ptr()),
              'ra_bn_moments_from_partials_f32')
      else:  # a channel count without epilogue moments (the one-channel output layer): float32 u, two-pass moments
        check(rn.lib().ra_conv3x3_bf16_f32(ptr(x), Cx, None, 0, B, Hs, Ws, int(stride == 2), ptr(wp), ptr(scale), ptr(shift), cout, 0, 1,
                                           ptr(u), None, 0, None, xin, rn.stream_ptr()), 'ra_conv3x3_bf16_f32')
        ws = _f(rn.lib().ra_bn_workspace_floats(cout), device=dev)
        check(rn.lib().ra_bn_moments_f32(ptr(u), B * H * W, cout, ptr(ws), ws.numel(), ptr(mean), ptr(var), rn.stream_ptr()),
              'ra_bn_moments_f32')
      y = alloc('y', (B, H // pool, W // pool, cout), yd)
      check(rn.lib().ra_bn_act_pool_bf16_f32(ptr(u), ptr(mean), ptr(var), ptr(gamma), ptr(beta), _C.c_float(BN_EPS), int(relu), int(pool),
                                             B, H, W, cout, ptr(y), (1 if st_ok else 0) | (2 if yd == torch.bfloat16 else 0),
                                             rn.stream_ptr()), 'ra_bn_act_pool_bf16_f32')
      return y, mean, var
    if use_bn and cout % 4 == 0 and EPILOGUE_MOMENTS['on']:
      # the batch moments ride on the conv epilogue: per-wave channel sums, then one small finishing launch
      u = place('u', B, Hs * up, Ws * up, cout)
      part = _f(rn.lib().ra_conv3x3_moments_part_floats(cout), device=dev)
      nparts = _C.c_int(0)
      check(rn.lib().ra_conv3x3_moments_f32(ptr(x), Cx, None, 0, B, Hs, Ws, int(stride == 2), ptr(wp), ptr(scale), ptr(shift),
                                            cout, 0, int(bf), ptr(u), ptr(part), part.numel(), _C.byref(nparts),
                                            rn.stream_ptr()), 'ra_conv3x3_moments_f32')
      check(rn.lib().ra_bn_moments_from_partials_f32(ptr(part), nparts.value, cout, ptr(mean), ptr(var), rn.stream_ptr()),
            'ra_bn_moments_from_partials_f32')
      H, W = u.shape[1], u.shape[2]
    else:
      u = ops.conv3x3(x, wp, scale, shift, cout, relu=False, pool=1, upsample=(stride == 2), bf16=bf,
                      out=place('u', B, Hs * up, Ws * up, cout))
      H, W = u.shape[1], u.shape[2]
      if use_bn:
        ws = _f(rn.lib().ra_bn_workspace_floats(cout), device=dev)
        check(rn.lib().ra_bn_moments_f32(ptr(u), B * H * W, cout, ptr(ws), ws.numel(), ptr(mean), ptr(var),
                                         rn.stream_ptr()), 'ra_bn_moments_f32')
    if use_bn:
      # whole-batch statistics under data parallelism (nnlib.py:98): one all_gather of (count, mean, var)
      ctx.n_total = sync_moments(mean, var, B * H * W) if meta.get('sync_bn') else 0.0
    y = place('y', B, H // pool, W // pool, cout)
    check(rn.lib().ra_bn_act_pool_f32(ptr(u), ptr(mean), ptr(var), ptr(gamma), ptr(beta), _C.c_float(BN_EPS), int(relu),
                                      int(pool), B, H, W, cout, ptr(y), rn.stream_ptr()), 'ra_bn_act_pool_f32')
    ctx.meta, ctx.cmap, ctx.cache = meta, cmap, cache
    if not use_bn:
      ctx.n_total = 0.0
    ctx.save_for_backward(x, w, u, mean, var, gamma, beta)
    if use_bn:
      ctx.mark_non_differentiable(mean, var)
      return y, mean, var
    z = torch.zeros(0, device=dev)
    ctx.mark_non_differentiable(z)
    return y, z, z

  @staticmethod
  def backward(ctx, dy, _dm, _dv):
    if dy is None:
      return None, None, None, None, None, None
    x, w, u, mean, var, gamma, beta = ctx.saved_tensors
    meta, cmap = ctx.meta, ctx.cmap
    dev = x.device
    tr, stride, pool, relu = meta['transposed'], meta['stride'], meta['pool'], meta['relu']
    B, Hs, Ws, Cx = x.shape
    _, H, W, cout = u.shape
    cin_w = w.shape[3] if tr else w.shape[2]
    dy = dy.contiguous()
    ws = _f(rn.lib().ra_bn_workspace_floats(cout), device=dev)
    dgamma, dbeta, du = _f(cout, device=dev), _f(cout, device=dev), torch.empty_like(u)
    grads = meta.get('grads')  # (gw, gb, ggamma, gbeta): views of the flat gradient bucket -> accumulate in-kernel
    deferred = grads is not None and meta.get('wgrad_defer') is not None  # the filter gradient waits for the end of backward
    nws = 0 if deferred else rn.lib().ra_conv3x3_wgrad_workspace_floats(Cx, cout, B, H, W)
    wws = None if deferred else _f(nws, device=dev)
    synced = ctx.n_total > 0.0 and ctx.n_total != float(B * H * W)
    bf = bool(meta.get('bf16'))  # compute_dtype = 'bf16': bf16 operands on the bf16 MFMA, float32 sums and tensors
    wgrad_acc = rn.lib().ra_conv3x3_wgrad_acc_bf16ops_f32 if bf else rn.lib().ra_conv3x3_wgrad_acc_f32
    wgrad_out = rn.lib().ra_conv3x3_wgrad_bf16ops_f32 if bf else rn.lib().ra_conv3x3_wgrad_f32

    def bn_backward(acc_g, acc_b):
      """dgamma / dbeta / du; with whole-batch statistics the two per-channel sums are all-reduced between the
      reduction and the du pass (the bucket still receives this rank's own sums: it is summed over the ranks later)."""
      if not synced:
        if acc_g is not None:
          check(rn.lib().ra_bn_act_pool_bwd_acc_f32(ptr(u), ptr(dy), ptr(mean), ptr(var), ptr(gamma), ptr(beta),
                                                    _C.c_float(BN_EPS), int(relu), int(pool), B, H, W, cout, ptr(ws), ws.numel(),
                                                    ptr(dgamma), ptr(dbeta), ptr(du), ptr(acc_g), ptr(acc_b), rn.stream_ptr()),
                'ra_bn_act_pool_bwd_acc_f32')
        else:
          check(rn.lib().ra_bn_act_pool_bwd_f32(ptr(u), ptr(dy), ptr(mean), ptr(var), ptr(gamma), ptr(beta),
                                                _C.c_float(BN_EPS), int(relu), int(pool), B, H, W, cout, ptr(ws), ws.numel(),
                                                ptr(dgamma), ptr(dbeta), ptr(du), rn.stream_ptr()), 'ra_bn_act_pool_bwd_f32')
        return dgamma, dbeta
      both = _f(2 * cout, device=dev)
      dgl, dbl = both[:cout], both[cout:]
      check(rn.lib().ra_bn_act_pool_bwd_reduce_f32(ptr(u), ptr(dy), ptr(mean), ptr(var), ptr(gamma), ptr(beta),
                                                   _C.c_float(BN_EPS), int(relu), int(pool), B, H, W, cout, ptr(ws), ws.numel(),
                                                   ptr(dgl), ptr(dbl), ptr(acc_g), ptr(acc_b), rn.stream_ptr()),
            'ra_bn_act_pool_bwd_reduce_f32')
      mine = both.clone()
      allreduce_sums(both)
      check(rn.lib().ra_bn_act_pool_bwd_dx_f32(ptr(u), ptr(dy), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(dgl), ptr(dbl),
                                               _C.c_double(ctx.n_total), _C.c_float(BN_EPS), int(relu), int(pool), B, H, W, cout,
                                               ptr(du), rn.stream_ptr()), 'ra_bn_act_pool_bwd_dx_f32')
      return mine[:cout], mine[cout:]  # the parameter gradients of THIS rank's shard (the bucket all-reduce sums them)

    if grads is not None:
      gw, gb, gg, gbt = grads
      bn_backward(gg, gbt)
      cmap_t = _const('cmap', tuple(cmap), dev, lambda: torch.tensor(cmap, dtype=torch.int32, device=dev)) \
          if cmap is not None else None
      if cmap is not None:
        real = [j for j in cmap if j >= 0]
        assert len(set(real)) == len(real), 'chan_map must be injective (one writer per gradient element)'
      defer = meta.get('wgrad_defer')
      if defer is not None:
        # the filter is shared by the step's timesteps: this call only adds its per-workgroup partial sums to the
        # layer's buffer; the end-of-backward callback (_DeferredWgrads) reduces each layer ONCE
        if not defer.armed:
          defer.armed = True
          torch.autograd.Variable._execution_engine.queue_callback(defer)
        slot = defer.get(meta['wgrad_key'])
        if slot is None:
          slot = defer[meta['wgrad_key']] = dict(calls=[], tab=torch.zeros(128, dtype=torch.int64, device=dev), ws=None)
        shape = (Cx, cout, B, Hs, Ws, H, W, int(stride == 2), tuple(cmap) if cmap is not None else None, int(cin_w), int(tr), bf)
        assert slot.setdefault('shape', shape) == shape, 'a deferred filter gradient needs one shape per layer'
        slot['calls'].append((x, du))  # nothing is launched here
        slot.update(cmap_t=cmap_t, gw=gw, gb=gb)
      else:
        check(wgrad_acc(ptr(x), Cx, B, Hs, Ws, int(stride == 2), ptr(du), cout, ptr(wws), nws,
                        ptr(cmap_t), int(cin_w), int(tr), ptr(gw), ptr(gb), rn.stream_ptr()), 'ra_conv3x3_wgrad_acc_f32')
      dw = db = None
      dgamma = dbeta = None
    else:
      dgamma, dbeta = bn_backward(None, None)
      # ---- backward-weight (of the SAME conv that ran) + bias
      dWf, db = _f(3, 3, Cx, cout, device=dev), _f(cout, device=dev)
      check(wgrad_out(ptr(x), Cx, B, Hs, Ws, int(stride == 2), ptr(du), cout, ptr(wws), nws,
                      ptr(dWf), ptr(db), rn.stream_ptr()), 'ra_conv3x3_wgrad_f32')
      if cmap is not None:  # packed kernel channels -> the filter's own input channels
        real = torch.zeros((3, 3, cin_w, cout), dtype=torch.float32, device=dev)
        for c, j in enumerate(cmap):
          if j >= 0:
            real[:, :, j, :] += dWf[:, :, c, :]
        dWf = real
      elif Cx != cin_w:
        dWf = dWf[:, :, :cin_w, :]
      dw = dWf.flip(0, 1).permute(0, 1, 3, 2).contiguous() if tr else dWf.contiguous()
    # ---- backward-data: the same MFMA conv kernel on the flipped / in-out-swapped packing
    dx = None
    if ctx.needs_input_grad[0]:
      duc = _pad_channels(du)
      cd = duc.shape[3]
      cpb = ops.cout_padded(cin_w)
      ones = _const('ones', cpb, dev, lambda: torch.ones(cpb, dtype=torch.float32, device=dev))
      zeros = _const('zeros', cpb, dev, lambda: torch.zeros(cpb, dtype=torch.float32, device=dev))
      wpb = _pack_dev(w.contiguous(), cout, cin_w, cd, _pad_map(cout, cd, dev), not tr, ctx.cache)
      dxr = ops.conv3x3(duc, wpb, ones, zeros, cin_w, relu=False, pool=1, bf16=bf)
      if stride == 2:
        sub = _f(B, Hs, Ws, cin_w, device=dev)
        check(rn.lib().ra_subsample_odd_f32(ptr(dxr), B, Hs, Ws, cin_w, ptr(sub), rn.stream_ptr()),
              'ra_subsample_odd_f32')
        dxr = sub
      if cmap is not None:
        dx = torch.zeros_like(x)
        for c, j in enumerate(cmap):
          if j >= 0:
            dx[..., c] = dxr[..., j]
      elif Cx != cin_w:
        dx = torch.zeros_like(x)
        dx[..., :cin_w] = dxr
      else:
        dx = dxr
    use_bn = gamma is not None
    return dx, dw, db, (dgamma if use_bn else None), (dbeta if use_bn else None), None


class BatchNormTrain(torch.autograd.Function):
  """nnlib.batch_norm with phase_train = True as an operator of its own (nnlib.py:98-119): tf.nn.moments over (B, H, W),
  gamma (x - mean) rsqrt(var + 1e-3) + beta, the batch statistics differentiated through.  x [B,H,W,C] float32.
  Returns (normed, batch mean, batch var); inside nnlib.cnn / dcnn the same kernels run fused behind the conv
  (ConvBNActPool)."""

  @staticmethod
  def forward(ctx, x, gamma, beta):
    ctx.set_materialize_grads(False)
    x = x.contiguous()
    B, H, W, C = x.shape
    dev = x.device
    mean, var = _f(C, device=dev), _f(C, device=dev)
    ws = _f(rn.lib().ra_bn_workspace_floats(C), device=dev)
    check(rn.lib().ra_bn_moments_f32(ptr(x), B * H * W, C, ptr(ws), ws.numel(), ptr(mean), ptr(var), rn.stream_ptr()),
          'ra_bn_moments_f32')
    y = torch.empty_like(x)
    check(rn.lib().ra_bn_act_pool_f32(ptr(x), ptr(mean), ptr(var), ptr(gamma), ptr(beta), _C.c_float(BN_EPS), 0, 1, B, H, W, C,
                                      ptr(y), rn.stream_ptr()), 'ra_bn_act_pool_f32')
    ctx.save_for_backward(x, mean, var, gamma, beta)
    ctx.mark_non_differentiable(mean, var)
    return y, mean, var

  @staticmethod
  def backward(ctx, dy, _dm, _dv):
    if dy is None:
      return None, None, None
    x, mean, var, gamma, beta = ctx.saved_tensors
    B, H, W, C = x.shape
    dev = x.device
    ws = _f(rn.lib().ra_bn_workspace_floats(C), device=dev)
    dgamma, dbeta, dx = _f(C, device=dev), _f(C, device=dev), torch.empty_like(x)
    check(rn.lib().ra_bn_act_pool_bwd_f32(ptr(x), ptr(dy.contiguous()), ptr(mean), ptr(var), ptr(gamma), ptr(beta),
                                          _C.c_float(BN_EPS), 0, 1, B, H, W, C, ptr(ws), ws.numel(), ptr(dgamma), ptr(dbeta),
                                          ptr(dx), rn.stream_ptr()), 'ra_bn_act_pool_bwd_f32')
    return dx, dgamma, dbeta


SPLIT_DGRAD = {'on': os.environ.get('RA_SPLIT_DGRAD', '1') != '0'}  # tuning aid: 0 = every float32 data gradient on K1


def _conv_dgrad(du, w, tr, stride, cmap, x_shape, cin_w, cache, bf, out_dtype=torch.float32):
  """Backward-data of ConvBNActPool's conv: the same MFMA conv kernel on the flipped / in-out-swapped packing.  In the bf16
  mode's stacked step du may be stored as bf16 and the result takes the input tensor's storage type (out_dtype)."""
  dev = du.device
  if du.dtype == torch.bfloat16 or out_dtype == torch.bfloat16:
    return _conv_dgrad_bf16(du, w, tr, stride, cmap, x_shape, cin_w, cache, out_dtype)
  B, Hs, Ws, Cx = x_shape
  cout = du.shape[3]
  cpb = ops.cout_padded(cin_w)
  ones = _const('ones', cpb, dev, lambda: torch.ones(cpb, dtype=torch.float32, device=dev))
  zeros = _const('zeros', cpb, dev, lambda: torch.zeros(cpb, dtype=torch.float32, device=dev))
  if (SPLIT_DGRAD['on'] and not bf and stride == 1 and cmap is None and Cx == cin_w and cout == 32 and  # (16 channels: conv16_kernel is as fast; 64: K1s's 8-row form is the decode loop's, not measured here)
      ops.conv_split_supported(cout, cin_w, 1, du.shape[1], du.shape[2]) and du.numel() * 4 < (1 << 31)):
    # round 5: the float32 data gradient of a 16- / 32-channel layer on the bf16 matrix pipe at float32 accuracy (K1s,
    # csrc/ra_conv_split.hip: every operand the exact sum of three bf16 pieces, six piece products) — 32 -> 32 at 128 x 128 x
    # 128 images: 2 x 292 -> 193 us in the cfg4 step.  The filter's pieces are packed on the device once per step (cache: per pack epoch).
    key = ('split', w.data_ptr(), w._version, bool(tr))
    hit = cache.get(key)
    if hit is None:
      hit = cache[key] = (w, ops.pack_split_weights_dev(w.detach(), cout, cin_w, transposed=not tr))
    return ops.conv_split(du.contiguous(), hit[1], ones, zeros, cin_w, relu=False, pool=1)
  duc = _pad_channels(du)
  cd = duc.shape[3]
  wpb = _pack_dev(w.contiguous(), cout, cin_w, cd, _pad_map(cout, cd, dev), not tr, cache)
  dxr = ops.conv3x3(duc, wpb, ones, zeros, cin_w, relu=False, pool=1, bf16=bf)
  if stride == 2:
    sub = _f(B, Hs, Ws, cin_w, device=dev)
    check(rn.lib().ra_subsample_odd_f32(ptr(dxr), B, Hs, Ws, cin_w, ptr(sub), rn.stream_ptr()), 'ra_subsample_odd_f32')
    dxr = sub
  if cmap is not None:
    dx = torch.zeros((B, Hs, Ws, Cx), dtype=torch.float32, device=dev)
    for c, j in enumerate(cmap):
      if j >= 0:
        dx[..., c] = dxr[..., j]
    return dx
  if Cx != cin_w:
    dx = torch.zeros((B, Hs, Ws, Cx), dtype=torch.float32, device=dev)
    dx[..., :cin_w] = dxr
    return dx
  return dxr


def _conv_dgrad_bf16(du, w, tr, stride, cmap, x_shape, cin_w, cache, out_dtype):
  dev = du.device
  B, Hs, Ws, Cx = x_shape
  cout = du.shape[3]
  assert cout % 4 == 0 or du.dtype == torch.float32  # bf16 du exists only behind the float4 BatchNorm kernels
  duc = _pad_channels(du)
  cd = duc.shape[3]
  cpb = ops.cout_padded(cin_w)
  ones = _const('ones', cpb, dev, lambda: torch.ones(cpb, dtype=torch.float32, device=dev))
  zeros = _const('zeros', cpb, dev, lambda: torch.zeros(cpb, dtype=torch.float32, device=dev))
  wpb = _pack_dev(w.contiguous(), cout, cin_w, cd, _pad_map(cout, cd, dev), not tr, cache)
  H, W = du.shape[1], du.shape[2]
  dxr = torch.empty((B, H, W, cin_w), dtype=out_dtype, device=dev)
  flags = (1 if du.dtype == torch.bfloat16 else 0) | (2 if out_dtype == torch.bfloat16 else 0)
  check(rn.lib().ra_conv3x3_bf16_f32(ptr(duc), cd, None, 0, B, H, W, 0, ptr(wpb), ptr(ones), ptr(zeros), cin_w, 0, 1, ptr(dxr), None, 0,
                                     None, flags, rn.stream_ptr()), 'ra_conv3x3_bf16_f32')
  if stride == 2:  # the transposed conv's stride: keep the odd positions (whole pixels: the element type does not matter)
    sub = torch.empty((B, Hs, Ws, cin_w), dtype=out_dtype, device=dev)
    if out_dtype == torch.bfloat16:
      assert cin_w % 2 == 0
      check(rn.lib().ra_subsample_odd_f32(ptr(dxr), B, Hs, Ws, cin_w // 2, ptr(sub), rn.stream_ptr()), 'ra_subsample_odd_f32')
    else:
      check(rn.lib().ra_subsample_odd_f32(ptr(dxr), B, Hs, Ws, cin_w, ptr(sub), rn.stream_ptr()), 'ra_subsample_odd_f32')
    dxr = sub
  if cmap is not None:
    dx = torch.zeros((B, Hs, Ws, Cx), dtype=out_dtype, device=dev)
    for c, j in enumerate(cmap):
      if j >= 0:
        dx[..., c] = dxr[..., j]
    return dx
  if Cx != cin_w:
    dx = torch.zeros((B, Hs, Ws, Cx), dtype=out_dtype, device=dev)
    dx[..., :cin_w] = dxr
    return dx
  return dxr


def stack_bn_reduce(info, U, dY):
  """Synchronised BatchNorm under the stacked backward, first half: the per-channel sums (sum dv, sum dv * xhat) of every
  timestep group of this rank's shard -> [G, 2 cout]; this rank's own sums also go to the gradient bucket (gamma / beta:
  the bucket all-reduce sums them over the ranks later)."""
  G, B, dev = info['G'], info['B'], U.device
  _, H, W, cout = U.shape
  ws = _f(rn.lib().ra_bn_workspace_floats(cout), device=dev)
  sums = _f(G, 2 * cout, device=dev)
  for g, (mean, var, gamma, beta, gg, gbt) in enumerate(info['per_group']):
    sl = slice(g * B, (g + 1) * B)
    check(rn.lib().ra_bn_act_pool_bwd_reduce_f32(ptr(U[sl]), ptr(dY[sl]), ptr(mean), ptr(var), ptr(gamma), ptr(beta),
                                                 _C.c_float(BN_EPS), int(info['relu']), int(info['pool']), B, H, W, cout, ptr(ws),
                                                 ws.numel(), ptr(sums[g, :cout]), ptr(sums[g, cout:]), ptr(gg), ptr(gbt),
                                                 rn.stream_ptr()), 'ra_bn_act_pool_bwd_reduce_f32')
  return sums


def stack_bn_dx(info, U, dY, sums, n_total):
  """... second half: du of every group from the sums of the WHOLE data-parallel batch (n_total elements per channel)."""
  G, B = info['G'], info['B']
  _, H, W, cout = U.shape
  du = torch.empty_like(U)
  for g, (mean, var, gamma, beta, _, _) in enumerate(info['per_group']):
    sl = slice(g * B, (g + 1) * B)
    check(rn.lib().ra_bn_act_pool_bwd_dx_f32(ptr(U[sl]), ptr(dY[sl]), ptr(mean), ptr(var), ptr(gamma), ptr(beta), ptr(sums[g, :cout]),
                                             ptr(sums[g, cout:]), _C.c_double(n_total), _C.c_float(BN_EPS), int(info['relu']),
                                             int(info['pool']), B, H, W, cout, ptr(du[sl]), rn.stream_ptr()), 'ra_bn_act_pool_bwd_dx_f32')
  return du


class ConvStackFn(torch.autograd.Function):
  """The T calls of one conv + BatchNorm + ReLU + pool layer of a training step (its T timesteps: shared filter,
  per-timestep statistics and BatchNorm parameters, nnlib.py:121-127) as ONE node of the autograd graph.

  The timesteps of full_model are coupled only through the canvas, whose gradient is stopped (full_model.py:843-848),
  and the controller state starts from zero in each: once the sequential forward has filled the layer's [T, ...] slabs
  (x, u, y, statistics), the backward passes of the T timesteps are independent and run here stacked along the batch —
  one grouped BatchNorm backward (ra_bn_act_pool_bwd_grouped_f32), one filter gradient and one data gradient over
  T * B images instead of T of each.  forward() returns the precomputed y slab."""

  @staticmethod
  def forward(ctx, X, w, b, info):
    ctx.set_materialize_grads(False)
    ctx.info = info
    ctx.save_for_backward(X, w)
    return info['Y']

  @staticmethod
  def backward(ctx, dY):
    if dY is None:
      return None, None, None, None
    X, w = ctx.saved_tensors
    info = ctx.info
    G, B, U, dev = info['G'], info['B'], info['U'], X.device
    tr, stride, pool, relu, cmap, bf = info['transposed'], info['stride'], info['pool'], info['relu'], info['chan_map'], info['bf16']
    N, Hs, Ws, Cx = X.shape
    _, H, W, cout = U.shape
    cin_w = w.shape[3] if tr else w.shape[2]
    gw, gb = info['gw'], info['gb']
    dY = dY.contiguous()
    nbn = rn.lib().ra_bn_workspace_floats(cout)
    rc = 0
    if info.get('sync_world', 1) > 1:
      # whole-batch BatchNorm (--sync_bn) under the stacked backward: the T groups' 2 C sums of a layer cross the ranks in
      # ONE all_reduce (21 collectives per step; the per-timestep graph issued T times as many)
      sums = stack_bn_reduce(info, U, dY)
      allreduce_sums(sums)
      du = stack_bn_dx(info, U, dY, sums, float(info['sync_world']) * B * H * W)
    else:
      du, rc = torch.empty_like(U), rn.RA_E_SHAPE
      sflags = (1 if U.dtype == torch.bfloat16 else 0) | (2 if dY.dtype == torch.bfloat16 else 0)  # the bf16 mode's storage
      if cout % 4 == 0 or not sflags:  # (a small one-channel layer: one workgroup per timestep of the same launch)
        dgam, dbet, ws = _f(G, cout, device=dev), _f(G, cout, device=dev), _f(G * nbn, device=dev)
        if sflags:
          rc = rn.lib().ra_bn_act_pool_bwd_grouped_bf16_f32(ptr(U), ptr(dY), ptr(info['tabs']), G, _C.c_float(BN_EPS), int(relu),
                                                            int(pool), B, H, W, cout, ptr(ws), ws.numel(), ptr(dgam), ptr(dbet), ptr(du),
                                                            sflags, rn.stream_ptr())
        else:
          rc = rn.lib().ra_bn_act_pool_bwd_grouped_f32(ptr(U), ptr(dY), ptr(info['tabs']), G, _C.c_float(BN_EPS), int(relu), int(pool),
                                                       B, H, W, cout, ptr(ws), ws.numel(), ptr(dgam), ptr(dbet), ptr(du), rn.stream_ptr())
        if rc != rn.RA_E_SHAPE or sflags:
          check(rc, 'ra_bn_act_pool_bwd_grouped_f32')
    if info.get('sync_world', 1) <= 1 and rc == rn.RA_E_SHAPE:
      # a channel count the grouped float4 kernel does not take (the one-channel output layer; C / 4 not a power of two,
      # e.g. the KITTI architecture's 96-channel layer): one call per timestep
      ws, dgam, dbet = _f(nbn, device=dev), _f(cout, device=dev), _f(cout, device=dev)
      for g, (mean, var, gamma, beta, gg, gbt) in enumerate(info['per_group']):
        sl = slice(g * B, (g + 1) * B)
        check(rn.lib().ra_bn_act_pool_bwd_acc_f32(ptr(U[sl]), ptr(dY[sl]), ptr(mean), ptr(var), ptr(gamma), ptr(beta),
                                                  _C.c_float(BN_EPS), int(relu), int(pool), B, H, W, cout, ptr(ws), ws.numel(),
                                                  ptr(dgam), ptr(dbet), ptr(du[sl]), ptr(gg), ptr(gbt), rn.stream_ptr()),
              'ra_bn_act_pool_bwd_acc_f32')
    cmap_t = _const('cmap', tuple(cmap), dev, lambda: torch.tensor(cmap, dtype=torch.int32, device=dev)) if cmap is not None else None
    nws = rn.lib().ra_conv3x3_wgrad_workspace_floats(Cx, cout, N, H, W)
    wws = info['cache'].get(('wgrad_ws', nws))
    if wws is None:
      wws = info['cache'][('wgrad_ws', nws)] = _f(nws, device=dev)
    wfmt = (1 if X.dtype == torch.bfloat16 else 0) | (2 if du.dtype == torch.bfloat16 else 0)
    if wfmt:
      check(rn.lib().ra_conv3x3_wgrad_acc_bf16_f32(ptr(X), Cx, N, Hs, Ws, int(stride == 2), ptr(du), cout, ptr(wws), nws, ptr(cmap_t),
                                                   int(cin_w), int(tr), ptr(gw), ptr(gb), wfmt, rn.stream_ptr()), 'ra_conv3x3_wgrad_acc_bf16_f32')
    else:
      wgrad = rn.lib().ra_conv3x3_wgrad_acc_bf16ops_f32 if bf else rn.lib().ra_conv3x3_wgrad_acc_f32
      check(wgrad(ptr(X), Cx, N, Hs, Ws, int(stride == 2), ptr(du), cout, ptr(wws), nws, ptr(cmap_t), int(cin_w), int(tr), ptr(gw),
                  ptr(gb), rn.stream_ptr()), 'ra_conv3x3_wgrad_acc_f32')
    dx = _conv_dgrad(du, w, tr, stride, cmap, X.shape, cin_w, info['cache'], bf, X.dtype) if ctx.needs_input_grad[0] else None
    return dx, None, None, None


class PairIoU(torch.autograd.Function):
  """modellib.f_iou(a, b, pairwise=True) (modellib.py:124-155) with its gradient in a: one
  streaming MFMA pass forward (K8), one weighted-sum pass backward.  tmajor: a (and its gradient) lie [N,B,H,W] — the
  stacked training step's masks, timestep-major as its sequential phase wrote them; read and written through strides."""

  @staticmethod
  def forward(ctx, a, b, tmajor=False):
    a, b = a.contiguous(), b.contiguous()
    st = ops.pair_stats(a, b, want=('iou_soft', 'inter', 'sum_a', 'sum_b'), a_tmajor=tmajor)
    ctx.save_for_backward(b, st['inter'], st['sum_a'], st['sum_b'])
    ctx.hw = a.shape[2] * a.shape[3]
    ctx.shape, ctx.tmajor = a.shape, bool(tmajor)
    return st['iou_soft']

  @staticmethod
  def backward(ctx, g):
    b, I, sa, sb = ctx.saved_tensors
    U = sa[:, :, None] + sb[:, None, :] - I + 1e-5 * ctx.hw
    c1 = (g * (U + I) / (U * U)).contiguous()     # d iou / d a_p = b_p (U + I) / U^2 - I / U^2
    c0 = (-(g * I / (U * U)).sum(dim=2)).contiguous()
    if ctx.tmajor:
      N, B, H, W = ctx.shape
      out = torch.empty((N, B, H, W), dtype=torch.float32, device=g.device)
      check(rn.lib().ra_weighted_sum_multi_strided_f32(ptr(c1), ptr(c0), ptr(b), B, N, b.shape[1], H * W, ptr(out), H * W, B * H * W,
                                                       rn.stream_ptr()), 'ra_weighted_sum_multi_strided_f32')
      return out, None, None
    B, N, H, W = ctx.shape
    out = torch.empty((B, N, H, W), dtype=torch.float32, device=g.device)
    check(rn.lib().ra_weighted_sum_multi_f32(ptr(c1), ptr(c0), ptr(b), B, N, b.shape[1], H * W, ptr(out),
                                             rn.stream_ptr()), 'ra_weighted_sum_multi_f32')
    return out, None, None


class LossHead(torch.autograd.Function):
  """The scalar head of the loss with the 'iou' box and mask losses (full_model.py:913-1035) as ONE launch each way
  (ra_loss_head_f32): matched soft IoUs, the confidence loss on the cumulative extrema of the scores, their mix; backward,
  the coefficient tensors of the two pairwise-IoU adjoints and d s_out in one launch, then the two weighted-sum passes that
  are the gradients of the masks and the boxes.  Replaces ~75 element-wise / reduction / scan launches of the autograd
  graph it stands for.  st_s / st_b: ops.pair_stats of (masks, y_gt) / (boxes, box_gt); m_s / m_b: their matchings."""

  @staticmethod
  def forward(ctx, y_out, attn_box, s_out, y_gt, box_gt, st_s, st_b, m_s, m_b, tmajor, mix):
    B, T = s_out.shape
    s_out = s_out.contiguous()
    buf = torch.empty(8, dtype=torch.float32, device=s_out.device)
    check(rn.lib().ra_loss_head_f32(ptr(st_s['iou_soft']), ptr(st_b['iou_soft']), ptr(m_s), ptr(m_b), ptr(s_out), B, T, float(mix), ptr(buf),
                                    rn.stream_ptr()), 'ra_loss_head_f32')
    ctx.save_for_backward(y_gt, box_gt, s_out, m_s, m_b, st_s['inter'], st_s['sum_a'], st_s['sum_b'], st_b['inter'], st_b['sum_a'],
                          st_b['sum_b'])
    ctx.meta = (bool(tmajor), float(mix), tuple(y_out.shape), tuple(attn_box.shape))
    pieces = buf[1:6]
    ctx.mark_non_differentiable(pieces)
    return buf[0], pieces   # loss; (box_loss, segm_loss, conf_loss, iou_soft, iou_soft_box)

  @staticmethod
  def backward(ctx, g, _gp):
    y_gt, box_gt, s_out, m_s, m_b, I_s, sa_s, sb_s, I_b, sa_b, sb_b = ctx.saved_tensors
    tm, mix, ysh, bsh = ctx.meta
    B, T = s_out.shape
    dev = s_out.device
    H, W = ysh[2], ysh[3]
    f = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    c1s, c0s, c1b, c0b, ds = f(B, T, T), f(B, T), f(B, T, T), f(B, T), f(B, T)
    g = g.contiguous()
    check(rn.lib().ra_loss_head_bwd_f32(ptr(g), ptr(m_s), ptr(m_b), ptr(s_out), ptr(I_s), ptr(sa_s), ptr(sb_s), ptr(I_b), ptr(sa_b),
                                        ptr(sb_b), B, T, H * W, mix, ptr(c1s), ptr(c0s), ptr(c1b), ptr(c0b), ptr(ds), rn.stream_ptr()),
          'ra_loss_head_bwd_f32')
    outs = []
    for c1, c0, gt, sh in ((c1s, c0s, y_gt, ysh), (c1b, c0b, box_gt, bsh)):
      out = torch.empty(sh, dtype=torch.float32, device=dev)
      if tm:   # [N,B,H,W]: written through strides
        check(rn.lib().ra_weighted_sum_multi_strided_f32(ptr(c1), ptr(c0), ptr(gt), B, T, gt.shape[1], H * W, ptr(out), H * W, B * H * W,
                                                         rn.stream_ptr()), 'ra_weighted_sum_multi_strided_f32')
      else:
        check(rn.lib().ra_weighted_sum_multi_f32(ptr(c1), ptr(c0), ptr(gt), B, T, gt.shape[1], H * W, ptr(out), rn.stream_ptr()),
              'ra_weighted_sum_multi_f32')
      outs.append(out)
    return outs[0], outs[1], ds, None, None, None, None, None, None, None, None


class LSTMCell(torch.autograd.Function):
  """Pointwise half of nnlib.lstm's cell (nnlib.py:641-646): pre [B,4*hid] in gate order (i, f, o, u)."""

  @staticmethod
  def forward(ctx, pre, c_prev):
    ctx.set_materialize_grads(False)  # the last cell of a timestep has no gradient in c: a null pointer, not zeros
    pre, c_prev = pre.contiguous(), c_prev.contiguous()
    B, hid = c_prev.shape
    h, c, act = torch.empty_like(c_prev), torch.empty_like(c_prev), torch.empty_like(pre)
    check(rn.lib().ra_lstm_cell_f32(ptr(pre), ptr(c_prev), B, hid, ptr(h), ptr(c), ptr(act), rn.stream_ptr()), 'ra_lstm_cell_f32')
    ctx.save_for_backward(act, c_prev, c)
    return h, c

  @staticmethod
  def backward(ctx, dh, dc):
    if dh is None and dc is None:
      return None, None
    act, c_prev, c = ctx.saved_tensors
    B, hid = c_prev.shape
    dpre, dcp = torch.empty_like(act), torch.empty_like(c_prev)
    dh = None if dh is None else dh.contiguous()
    dc = None if dc is None else dc.contiguous()
    check(rn.lib().ra_lstm_cell_bwd_f32(ptr(act), ptr(c_prev), ptr(c), ptr(dh), ptr(dc), B, hid, ptr(dpre), ptr(dcp),
                                        rn.stream_ptr()), 'ra_lstm_cell_bwd_f32')
    return dpre, dcp


class GaussFilter(torch.autograd.Function):
  """modellib.get_gaussian_filter (modellib.py:581-612) with its adjoint: ctr, size, lg_var [B] -> [B,L,F]."""

  @staticmethod
  def forward(ctx, ctr, size, lg_var, L, F):
    ctr, size, lg_var = ctr.contiguous(), size.contiguous(), lg_var.contiguous()
    B = ctr.shape[0]
    out = torch.empty((B, L, F), dtype=torch.float32, device=ctr.device)
    check(rn.lib().ra_gauss_filter_f32(ptr(ctr), ptr(size), ptr(lg_var), B, int(L), int(F), ptr(out), rn.stream_ptr()),
          'ra_gauss_filter_f32')
    ctx.save_for_backward(ctr, size, lg_var)
    ctx.dims = (int(L), int(F))
    return out

  @staticmethod
  def backward(ctx, g):
    ctr, size, lg_var = ctx.saved_tensors
    L, F = ctx.dims
    B = ctr.shape[0]
    dc, ds, dv = torch.empty_like(ctr), torch.empty_like(ctr), torch.empty_like(ctr)
    check(rn.lib().ra_gauss_filter_bwd_f32(ptr(ctr), ptr(size), ptr(lg_var), ptr(g.contiguous()), B, L, F, ptr(dc), ptr(ds),
                                           ptr(dv), rn.stream_ptr()), 'ra_gauss_filter_bwd_f32')
    return dc, ds, dv, None, None


def gaussian_filter(ctr, size, lg_var, L, F):
  """modellib.get_gaussian_filter (modellib.py:581-612), differentiable: ctr, size, lg_var [B]."""
  return GaussFilter.apply(ctr, size, lg_var, L, F)


class LinearAcc(torch.autograd.Function):
  """y = x W + b whose parameter gradients are ADDED in place to gw / gb (views of the trainer's flat gradient bucket,
  or a step-local buffer): one GEMM (beta = 1) and one GEMV per use, where autograd's mm + sum + the `grad += g`
  that merges the uses of a shared weight (80 uses of the LSTM's, 64 of each glimpse-MLP layer's) are four launches."""

  @staticmethod
  def forward(ctx, x, W, b, gw, gb, after=None):
    """after: a callable run ONCE at the end of the backward pass this use takes part in (the packed LSTM weights'
    buffers are scattered to the eight parameters' gradients there)."""
    ctx.save_for_backward(x, W)
    ctx.acc = (gw, gb, after)
    return torch.addmm(b, x, W)

  @staticmethod
  def backward(ctx, dy):
    x, W = ctx.saved_tensors
    gw, gb, after = ctx.acc
    if after is not None and not after.armed:
      after.armed = True
      torch.autograd.Variable._execution_engine.queue_callback(after)
    gw.addmm_(x.t(), dy)
    ones = _const('ones', dy.shape[0], dy.device, lambda: torch.ones(dy.shape[0], dtype=torch.float32, device=dy.device))
    gb.addmv_(dy.t(), ones)
    return (dy @ W.t() if ctx.needs_input_grad[0] else None), None, None, None, None, None


class _LstmGradScatter(object):
  """End-of-backward callback: the step-local gradient of the packed gate weights goes to the eight parameters."""

  def __init__(self, trainer, gW, gb):
    self.trainer, self.gW, self.gb, self.armed = trainer, gW, gb, False

  def __call__(self):
    self.armed = False
    g, hid = self.trainer.bucket.grad_of, self.trainer.d['hid']
    nx = g['ctrl_lstm_w_xi'].shape[0]
    for j, k in enumerate('ifou'):
      cols = slice(j * hid, (j + 1) * hid)
      g['ctrl_lstm_w_x' + k].add_(self.gW[:nx, cols])
      g['ctrl_lstm_w_h' + k].add_(self.gW[nx:, cols])
      g['ctrl_lstm_b_' + k].add_(self.gb[cols])
    self.gW.zero_()  # a second backward through the same forward must not count this one again
    self.gb.zero_()


def _dense(g):
  return None if g is None else g.contiguous()


class _CtrlStepBuffers(object):
  """Per-trainer buffers of the fused controller (csrc/ra_ctrl_train.hip): what the forward of each timestep saves,
  the pre-activation gradients the backward writes, and the end-of-backward pass that turns them into the parameter
  gradients — the layer inputs of ALL images, glimpse iterations and timesteps of the step against their
  pre-activation gradients: four GEMMs + four bias sums per optimisation step (the library path issued 960)."""

  def __init__(self, trainer, T, B):
    d, dev = trainer.d, trainer.bucket.param.device
    self.trainer, self.T, self.B, self.armed = trainer, T, B, False
    self.G, self.Cf, self.hid, self.iters = d['G'], trainer.model.dims['ccnn_channels'][-1], d['hid'], d['iters']
    self.n_g, self.n_c, self.mlp = int(d['n_gmlp']), int(d['n_cmlp']), int(d.get('mlp_dim', 0) or 0)
    self.SF = rn.lib().ra_ctrl_train_save_floats_n(self.G, self.Cf, self.hid, self.iters, self.n_g) // self.iters
    f = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)
    self.save = f(T, B, self.iters, self.SF)
    self.dpre, self.dlog = f(T, B, self.iters, 4 * self.hid), f(T, B, self.iters, self.G)
    # pre-activation gradients of the MLPs' hidden layers, layer-major: glimpse [n_g - 1][T, B, iters, hid], controller
    # [n_c - 1][T, B, mlp]; what the controller MLP's hidden layers put out [T, B, (n_c - 1) mlp]
    self.dzg = f(max(self.n_g - 1, 1), T, B, self.iters, self.hid)
    self.dz1 = self.dzg[0]
    self.dzc = f(max(self.n_c - 1, 1), T, B, max(self.mlp, 1))
    self.save_c = f(T, B, max((self.n_c - 1) * self.mlp, 1))
    self.hfin, self.dco = f(T, B, self.hid), f(T, B, 9)
    self.gW = f(self.Cf + self.hid, 4 * self.hid)

  def begin_step(self):
    for t in (self.dpre, self.dzg, self.dzc, self.dlog, self.dco):  # a timestep whose backward does not run must not leave last step's rows
      t.zero_()

  def __call__(self):  # end of the backward pass: the parameter gradients, into the bucket
    self.armed = False
    g, hid, Cf, it = self.trainer.bucket.grad_of, self.hid, self.Cf, self.iters
    n = self.T * self.B * it
    SF, G = self.SF, self.G
    n_g, n_c, mlp, TB = self.n_g, self.n_c, self.mlp, self.T * self.B
    if Cf % 16 == 0 and hid % 16 == 0 and getattr(self.trainer, 'own_gemm', True):
      # one launch of the library's short-K GEMM per dense layer (csrc/ra_gemm.hip): the bias gradients ride along as one
      # more output row, and the LSTM's product lands directly in its eight weight / four bias tensors through a pointer table
      if getattr(self, '_seg', None) is None:
        self._seg = torch.tensor([g['ctrl_lstm_w_x' + k].data_ptr() for k in 'ifou'] + [g['ctrl_lstm_w_h' + k].data_ptr() for k in 'ifou'] +
                                 [g['ctrl_lstm_b_' + k].data_ptr() for k in 'ifou'], dtype=torch.int64, device=self.save.device)
      base, f4 = self.save.data_ptr(), 4
      gemm = lambda *a: check(rn.lib().ra_gemm_tn_acc_f32(*a, rn.stream_ptr()), 'ra_gemm_tn_acc_f32')
      gemm(base, SF, ptr(self.dpre), 4 * hid, n, Cf + hid, 4 * hid, 0, None, 0, None, ptr(self._seg), Cf, hid)
      zcol = lambda l: base + f4 * (Cf + 6 * hid + l * hid)     # hidden layer l of the glimpse MLP inside a saved row
      last_w, last_b = g['glimpse_mlp_w_%d' % (n_g - 1)], g['glimpse_mlp_b_%d' % (n_g - 1)]
      if n_g == 1:   # logits = h W + b: h after the LSTM of (b, it) = the h half of iteration it + 1's LSTM input
        gemm(base + f4 * (SF + Cf), SF, ptr(self.dlog), G, n - 1, hid, G, it, ptr(last_w), G, ptr(last_b), None, 0, 0)
      else:
        # layer 0 reads that h: A one row ahead of B, the last iteration of every image skipped
        gemm(base + f4 * (SF + Cf), SF, ptr(self.dzg[0]), hid, n - 1, hid, hid, it, ptr(g['glimpse_mlp_w_0']), hid, ptr(g['glimpse_mlp_b_0']),
             None, 0, 0)
        for l in range(1, n_g - 1):  # hidden layer l reads hidden layer l - 1 of the same (image, iteration)
          gemm(zcol(l - 1), SF, ptr(self.dzg[l]), hid, n, hid, hid, 0, ptr(g['glimpse_mlp_w_%d' % l]), hid, ptr(g['glimpse_mlp_b_%d' % l]),
               None, 0, 0)
        gemm(zcol(n_g - 2), SF, ptr(self.dlog), G, n, hid, G, 0, ptr(last_w), G, ptr(last_b), None, 0, 0)
      for l in range(n_c):  # controller MLP: [hid] + [mlp] * (n_c - 1) + [9]
        A, lda, K = (ptr(self.hfin), hid, hid) if l == 0 else (self.save_c.data_ptr() + f4 * (l - 1) * mlp, (n_c - 1) * mlp, mlp)
        Bm, N = (ptr(self.dco), 9) if l == n_c - 1 else (ptr(self.dzc[l]), mlp)
        gemm(A, lda, Bm, N, TB, K, N, 0, ptr(g['ctrl_mlp_w_%d' % l]), N, ptr(g['ctrl_mlp_b_%d' % l]), None, 0, 0)
      return
    rows = self.save.view(n, self.SF)
    ones = _const('ones', n, rows.device, lambda: torch.ones(n, dtype=torch.float32, device=rows.device))
    D = self.dpre.view(n, 4 * hid)
    torch.mm(rows[:, :Cf + hid].t(), D, out=self.gW)
    nx = Cf
    for j, k in enumerate('ifou'):
      cols = slice(j * hid, (j + 1) * hid)
      g['ctrl_lstm_w_x' + k].add_(self.gW[:nx, cols])
      g['ctrl_lstm_w_h' + k].add_(self.gW[nx:, cols])
      g['ctrl_lstm_b_' + k].addmv_(D[:, cols].t(), ones)
    # the glimpse MLP's first layer reads h after the LSTM of (b, it) = the h half of iteration it + 1's LSTM input
    sv4 = self.save.view(self.T, self.B, it, self.SF)
    H0 = sv4[:, :, 1:, Cf:Cf + hid].reshape(-1, hid)
    DL = self.dlog.view(n, self.G)
    for l in range(n_g):
      A = H0 if l == 0 else rows[:, Cf + 6 * hid + (l - 1) * hid:Cf + 6 * hid + l * hid]
      Bm = DL if l == n_g - 1 else self.dzg[l].reshape(n, hid)
      if l == 0:
        Bm = Bm.view(self.T, self.B, it, -1)[:, :, :-1].reshape(-1, Bm.shape[1])
      g['glimpse_mlp_w_%d' % l].addmm_(A.t(), Bm)
      g['glimpse_mlp_b_%d' % l].addmv_(Bm.t(), ones[:Bm.shape[0]])
    HF = self.hfin.view(-1, hid)
    for l in range(n_c):
      A = HF if l == 0 else self.save_c.view(TB, -1)[:, (l - 1) * mlp:l * mlp]
      Bm = self.dco.view(-1, 9) if l == n_c - 1 else self.dzc[l].reshape(TB, mlp)
      g['ctrl_mlp_w_%d' % l].addmm_(A.t(), Bm)
      g['ctrl_mlp_b_%d' % l].addmv_(Bm.t(), ones[:TB])


class ControllerFn(torch.autograd.Function):
  """full_model.py:668-689 for one timestep: feat [B,G,Cf] -> (h_last [B,hid], ctrl_out [B,9]) in ONE launch, its
  adjoint (BPTT over the glimpse iterations) in one more; any depth of the glimpse and controller MLPs (up to 4 layers
  each).  The weights are passed as plain tensors (gws = the glimpse MLP's [(w, b), ...], cws = the controller MLP's): their
  gradients are formed once per step by the trainer's _CtrlStepBuffers from the rows this function saves."""

  @staticmethod
  def forward(ctx, feat, Wg, bg, gws, cws, bufs, tt):
    ctx.set_materialize_grads(False)
    feat = feat.contiguous()
    B, G, Cf = feat.shape
    hid, iters = bufs.hid, bufs.iters
    sel = (lambda t: t.view((-1,) + t.shape[2:])) if tt == 'all' else (lambda t: t[tt])  # 'all': the T timesteps stacked
    h, co = sel(bufs.hfin), torch.empty((B, 9), dtype=torch.float32, device=feat.device)
    arr = lambda ts: (_C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    check(rn.lib().ra_ctrl_train_fwd_n_f32(B, G, Cf, hid, iters, 9, bufs.n_g, bufs.n_c, bufs.mlp, ptr(feat), ptr(Wg), ptr(bg),
                                           arr([w for w, _ in gws]), arr([b for _, b in gws]), arr([w for w, _ in cws]),
                                           arr([b for _, b in cws]), ptr(h), ptr(co), ptr(sel(bufs.save)),
                                           ptr(sel(bufs.save_c)) if bufs.n_c > 1 else None, rn.stream_ptr()), 'ra_ctrl_train_fwd_n_f32')
    ctx.save_for_backward(feat, Wg, *([w for w, _ in gws] + [w for w, _ in cws]))
    ctx.bufs, ctx.tt = bufs, tt
    return h, co  # h is the trainer's row buffer hfin[tt] (the GEMM input of the controller MLP's gradient): nothing writes it again this step

  @staticmethod
  def backward(ctx, dh, dco):
    feat, Wg = ctx.saved_tensors[:2]
    bufs, tt = ctx.bufs, ctx.tt
    gW, cW = ctx.saved_tensors[2:2 + bufs.n_g], ctx.saved_tensors[2 + bufs.n_g:]
    if not bufs.armed:
      bufs.armed = True
      torch.autograd.Variable._execution_engine.queue_callback(bufs)
    B, G, Cf = feat.shape
    dfeat = torch.empty_like(feat)
    sel = (lambda t: t.view((-1,) + t.shape[2:])) if tt == 'all' else (lambda t: t[tt])
    lsel = (lambda t: t.view((t.shape[0], -1) + t.shape[3:])) if tt == 'all' else (lambda t: t[:, tt])  # layer-major buffers
    if dco is not None:
      sel(bufs.dco).copy_(dco)
    arr = lambda ts: (_C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    dzg, dzc = lsel(bufs.dzg), lsel(bufs.dzc)
    check(rn.lib().ra_ctrl_train_bwd_n_f32(B, G, Cf, bufs.hid, bufs.iters, 9, bufs.n_g, bufs.n_c, bufs.mlp, ptr(feat), ptr(Wg), arr(gW), arr(cW),
                                           ptr(sel(bufs.save)), ptr(sel(bufs.save_c)) if bufs.n_c > 1 else None, ptr(_dense(dh)),
                                           ptr(sel(bufs.dco)) if dco is not None else None, ptr(dfeat), ptr(sel(bufs.dpre)),
                                           ptr(sel(bufs.dlog)), ptr(dzg) if bufs.n_g > 1 else None, int(dzg.stride(0)),
                                           ptr(dzc) if bufs.n_c > 1 else None, int(dzc.stride(0)), rn.stream_ptr()), 'ra_ctrl_train_bwd_n_f32')
    return (dfeat,) + (None,) * 6


class AttnHead(torch.autograd.Function):
  """Controller output [B,>=9] -> (cn, ls, ctr, size, lg_var [B,2], attn_gamma, box_gamma, y_lg_gamma [B]) of
  full_model.py:702-722 / modellib.py:752-764,812-825: one launch forward, one backward (ra_attn_head_f32)."""

  @staticmethod
  def forward(ctx, co, H, W, Fh, Fw, flags):
    ctx.set_materialize_grads(False)
    co = co if (co.dim() == 2 and co.stride(1) == 1) else co.contiguous()
    B = co.shape[0]
    rec = torch.empty((B, 16), dtype=torch.float32, device=co.device)
    # arec: the same window as the resample kernels' attention record (every kernel reads only its own gamma column, so
    # one record serves the box, the extract and the paste: no torch.cat per use)
    arec = torch.empty((B, rn.RA_ATTN_STRIDE), dtype=torch.float32, device=co.device)
    check(rn.lib().ra_attn_head_rec_f32(ptr(co), int(co.stride(0)), B, int(H), int(W), int(Fh), int(Fw), int(flags), ptr(rec),
                                        ptr(arec), rn.stream_ptr()), 'ra_attn_head_rec_f32')
    ctx.save_for_backward(co, rec)
    ctx.meta = (int(H), int(W), int(flags))
    ctx.mark_non_differentiable(arec)
    return rec[:, 0:2], rec[:, 2:4], rec[:, 4:6], rec[:, 6:8], rec[:, 8:10], rec[:, 10], rec[:, 11], rec[:, 12], arec

  @staticmethod
  def backward(ctx, *gs):
    co, rec = ctx.saved_tensors
    H, W, flags = ctx.meta
    B = co.shape[0]
    gs = [_dense(g) for g in gs[:8]]
    dco = torch.empty((B, co.shape[1]), dtype=torch.float32, device=co.device)
    if co.shape[1] != 9:
      dco.zero_()
    out9 = dco if co.shape[1] == 9 else torch.empty((B, 9), dtype=torch.float32, device=co.device)
    check(rn.lib().ra_attn_head_bwd_f32(ptr(co), int(co.stride(0)), ptr(rec), *[ptr(g) for g in gs], B, H, W, flags, ptr(out9),
                                        rn.stream_ptr()), 'ra_attn_head_bwd_f32')
    if out9 is not dco:
      dco[:, :9] = out9
    return dco, None, None, None, None, None


class KnobMix(torch.autograd.Function):
  """The ground-truth knob on the attention window (full_model.py:744-773): (ctr, size) <- knob * matched noisy GT box
  + (1 - knob) * prediction, the matched box being sum_t match[b][t] gt[b][t] — one launch each way."""

  @staticmethod
  def forward(ctx, ctr, size, match, ctr_gt, size_gt, knob, arec=None):
    """arec: the head's attention record; the third output is that record with the mixed window (None without it)."""
    ctx.set_materialize_grads(False)
    B, T = match.shape
    if not (ctr.stride(1) == 1 and size.stride(1) == 1 and ctr.stride(0) == size.stride(0)):
      ctr, size = ctr.contiguous(), size.contiguous()
    knob = knob.reshape(B, -1)[:, 0]
    ctr2 = torch.empty((B, 2), dtype=torch.float32, device=ctr.device)
    size2 = torch.empty_like(ctr2)
    arec2 = None if arec is None else torch.empty_like(arec)
    check(rn.lib().ra_knob_mix_rec_f32(ptr(ctr), ptr(size), ptr(match.contiguous()), ptr(ctr_gt.contiguous()), ptr(size_gt.contiguous()),
                                       ptr(knob), int(knob.stride(0)), int(ctr.stride(0)), B, T, ptr(ctr2), ptr(size2), ptr(arec),
                                       ptr(arec2), rn.stream_ptr()), 'ra_knob_mix_rec_f32')
    ctx.save_for_backward(knob)
    if arec2 is not None:
      ctx.mark_non_differentiable(arec2)
    return ctr2, size2, arec2

  @staticmethod
  def backward(ctx, g_ctr, g_size, _g_rec=None):
    knob, = ctx.saved_tensors
    B = knob.shape[0]
    d_ctr = torch.empty((B, 2), dtype=torch.float32, device=knob.device)
    d_size = torch.empty_like(d_ctr)
    check(rn.lib().ra_knob_mix_bwd_f32(ptr(_dense(g_ctr)), ptr(_dense(g_size)), ptr(knob), int(knob.stride(0)), B, ptr(d_ctr),
                                       ptr(d_size), rn.stream_ptr()), 'ra_knob_mix_bwd_f32')
    return d_ctr, d_size, None, None, None, None, None


class GaussFilterPair(torch.autograd.Function):
  """Both banks of an attention window from the [B,2] tensors (y, x) the controller head produces: two kernel
  launches forward and two backward, reading the columns where they lie and writing the [B,2] gradients in place
  — per timestep this replaced 12 column copies, 6 zero-fills + 6 scatters of select_backward and the adds that
  merged them."""

  @staticmethod
  def forward(ctx, ctr, size, lg_var, H, W, Fh, Fw):
    ts = [t if (t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= 1) else t.contiguous() for t in (ctr, size, lg_var)]
    B, dev = ctr.shape[0], ctr.device
    outs = []
    for k, (L, F) in enumerate(((H, Fh), (W, Fw))):
      out = torch.empty((B, int(L), int(F)), dtype=torch.float32, device=dev)
      check(rn.lib().ra_gauss_filter_strided_f32(*[_C.c_void_p(t.data_ptr() + 4 * k) for t in ts], *[int(t.stride(0)) for t in ts],
                                                 B, int(L), int(F), ptr(out), rn.stream_ptr()), 'ra_gauss_filter_strided_f32')
      outs.append(out)
    ctx.save_for_backward(*ts)
    ctx.dims = ((int(H), int(Fh)), (int(W), int(Fw)))
    return outs[0], outs[1]

  @staticmethod
  def backward(ctx, gy, gx):
    ts = ctx.saved_tensors
    B, dev = ts[0].shape[0], ts[0].device
    make = torch.zeros if (gy is None or gx is None) else torch.empty
    grads = [make((B, 2), dtype=torch.float32, device=dev) for _ in range(3)]
    for k, (g, (L, F)) in enumerate(zip((gy, gx), ctx.dims)):
      if g is None:
        continue
      check(rn.lib().ra_gauss_filter_strided_bwd_f32(*[_C.c_void_p(t.data_ptr() + 4 * k) for t in ts],
                                                     *[int(t.stride(0)) for t in ts], ptr(g.contiguous()), B, L, F,
                                                     *[_C.c_void_p(t.data_ptr() + 4 * k) for t in grads], 2, rn.stream_ptr()),
            'ra_gauss_filter_strided_bwd_f32')
    return grads[0], grads[1], grads[2], None, None, None, None


def gaussian_filters(ctr, size, lg_var, H, W, Fh, Fw):
  """(F_y [B,H,Fh], F_x [B,W,Fw]) of modellib.get_gaussian_filter for ctr, size, lg_var [B,2] = (y, x)."""
  return GaussFilterPair.apply(ctr, size, lg_var, H, W, Fh, Fw)


def attn_record(ctr, size, lg_var, attn_gamma=None, box_gamma=None, y_lg_gamma=None):
  """The [B, RA_ATTN_STRIDE] attention record the resample kernels read (include/recattend.h: 0 ctr_y, 1 ctr_x,
  2 size_y, 3 size_x, 4 lg_var_y, 5 lg_var_x, 6 attn_gamma, 7 box_gamma, 8 y_lg_gamma) from the training graph's
  [B,2] / [B] tensors: one launch."""
  B, dev = ctr.shape[0], ctr.device
  one = _const('ones', (B, 1), dev, lambda: torch.ones((B, 1), dtype=torch.float32, device=dev))
  pad = _const('zeros', (B, rn.RA_ATTN_STRIDE - 9), dev, lambda: torch.zeros((B, rn.RA_ATTN_STRIDE - 9), dtype=torch.float32, device=dev))
  col = lambda g, dflt: dflt if g is None else g.detach().reshape(B, 1)
  return torch.cat([ctr.detach(), size.detach(), lg_var.detach(), col(attn_gamma, one), col(box_gamma, one),
                    col(y_lg_gamma, pad[:, :1]), pad], dim=1)


class AttnExtract(torch.autograd.Function):
  """x_patch = attn_gamma * F_y^T X F_x (modellib.extract_patch, full_model.py:778-789) on the banded HIP kernels: the
  forward is the decode loop's extract (weights evaluated on the fly from the window parameters), the backward turns
  d x_patch straight into d (ctr, size, lg_var, attn_gamma) — no [L,F] filter banks, no GEMMs.  X is differentiated only
  where the caller asks (stop_canvas_grad = False, full_model.py:843-848: the canvas channel of X carries a gradient):
  d X = gamma F_y dP F_x^T, the reference's own "paste" (extract_patch with the transposed banks, modellib.py:615-641)
  on the dense-bank operator."""

  @staticmethod
  def forward(ctx, x, ctr, size, lg_var, gamma, Fh, Fw, out=None, pre=None, rec=None):
    """out: where the patch is written (a [T, ...] slab's slice); pre: the patch, computed before (the stacked step's
    sequential phase ran this extract timestep by timestep: its graph node only needs the backward); rec: the attention
    record of (ctr, size, lg_var, gamma) where the caller has it ([record], AttnHead / KnobMix write one)."""
    ctx.set_materialize_grads(False)
    x = x.contiguous()
    B, H, W, C = x.shape
    rec = attn_record(ctr, size, lg_var, attn_gamma=gamma) if rec is None else rec[0]
    if pre is not None:
      patch = pre[0]   # (wrapped in a list: a tensor argument returned as the output would be seen as an in-place pass-through)
    else:
      patch = out if out is not None else torch.empty((B, Fh, Fw, C), dtype=torch.float32, device=x.device)
      ops.extract_direct(x, 0, rec, Fh, Fw, C, True, patch)
    ctx.save_for_backward(x, rec)
    ctx.dims = (H, W, int(Fh), int(Fw))
    return patch

  @staticmethod
  def backward(ctx, g):
    if g is None:
      return (None,) * 10
    x, rec = ctx.saved_tensors
    H, W, Fh, Fw = ctx.dims
    out = ops.resample_bwd(ops.RESAMPLE_READ, rec, H, W, Fh, Fw, X=x, Q=g.contiguous(), scale=rec[:, 6])
    dx = None
    if ctx.needs_input_grad[0]:
      col = lambda k: rec[:, k].contiguous()
      fyT = ops.gaussian_filter(col(0), col(2), col(4), H, Fh).transpose(1, 2).contiguous()   # [B,Fh,H]
      fxT = ops.gaussian_filter(col(1), col(3), col(5), W, Fw).transpose(1, 2).contiguous()   # [B,Fw,W]
      dx = ops.extract_patch_dense((g * rec[:, 6].view(-1, 1, 1, 1)).contiguous(), fyT, fxT)    # [B,H,W,C]
    return dx, out[:, 0:2], out[:, 2:4], out[:, 4:6], out[:, 6], None, None, None, None, None


class AttnPaste(torch.autograd.Function):
  """y = sigmoid(gain * F_y P F_x^T - 5) [B,H,W] (full_model.py:810-818 with gain = exp(y_lg_gamma); the attention box
  :738-741 with P == 1 and gain = box_gamma when `patch` is None) and its adjoint: d patch, d (ctr, size, lg_var) and the
  gamma gradient from one banded launch."""

  @staticmethod
  def forward(ctx, patch, ctr, size, lg_var, gamma, H, W, Fh, Fw, out=None, pre=None, rec=None):
    """out / pre / rec: as AttnExtract's (the plane(s) to write into; the planes computed before; the attention record)."""
    ctx.set_materialize_grads(False)
    B, dev = ctr.shape[0], ctr.device
    box = patch is None
    if rec is not None:
      rec = rec[0]
    else:
      rec = attn_record(ctr, size, lg_var, box_gamma=gamma) if box else attn_record(ctr, size, lg_var, y_lg_gamma=gamma)
    y = pre[0] if pre is not None else (out if out is not None else torch.empty((B, H, W), dtype=torch.float32, device=dev))
    if box:
      if pre is None:
        ops.attn_box_direct(rec, H, W, Fh, Fw, -5.0, y, H * W)
      ctx.save_for_backward(rec, y)
    else:
      patch = patch.reshape(B, Fh, Fw, 1).contiguous()
      if pre is None:
        ops.paste_direct(patch, 0, rec, -5.0, False, y, H * W, H, W)
      ctx.save_for_backward(rec, y, patch)
    ctx.dims, ctx.box = (int(H), int(W), int(Fh), int(Fw)), box
    return y

  @staticmethod
  def backward(ctx, g):
    if g is None:
      return (None,) * 12
    H, W, Fh, Fw = ctx.dims
    if ctx.box:
      rec, y = ctx.saved_tensors
      out = ops.resample_bwd(ops.RESAMPLE_BOX, rec, H, W, Fh, Fw, dY=g.contiguous(), Y=y, div=rec[:, 7])
      dp = None
    else:
      rec, y, patch = ctx.saved_tensors
      dp = torch.empty_like(patch)
      out = ops.resample_bwd(ops.RESAMPLE_WRITE, rec, H, W, Fh, Fw, dY=g.contiguous(), Y=y, Q=patch, E=dp)
    return dp, out[:, 0:2], out[:, 2:4], out[:, 4:6], out[:, 6], None, None, None, None, None, None, None


def _flat_bn_statistics(trainer):
  """One flat buffer for every BatchNorm shadow (ema_mean | ema_var, per layer and timestep copy) and a
  buffer of the same layout that the moment kernels write the batch statistics into: the EMA update
  of a step (nnlib.py:103-110, 2 x 640 small tensors at the CVPPP arch) becomes two launches."""
  model, dev = trainer.model, trainer.bucket.param.device
  keys = sorted(k[:-len('_ema_mean')] for k in model.weight_keys() if k.endswith('_ema_mean'))
  offs, off = {}, 0
  for k in keys:
    n = model[k + '_ema_mean'].numel()
    offs[k] = (off, n)
    off += n
  trainer.ema = torch.zeros(2 * off, dtype=torch.float32, device=dev)
  trainer.stat = torch.zeros(2 * off, dtype=torch.float32, device=dev)
  trainer._stat_views = {}
  for k, (o, n) in offs.items():
    for half, name in ((0, '_ema_mean'), (off, '_ema_var')):
      view = trainer.ema[half + o:half + o + n]
      view.copy_(model[k + name].reshape(-1).to(dev))
      model[k + name] = view.view(model[k + name].shape)  # the model (and the decode engine) read the flat buffer
    trainer._stat_views[k] = (trainer.stat[o:o + n], trainer.stat[off + o:off + o + n])
  eng = getattr(model, 'engine', None)
  if eng is not None:
    eng._stamp = None


class TrainStep(object):
  """model.run(['loss', 'train_step'], feed) of the reference's trainer (full_model_train.py:107)."""

  def __init__(self, model, world=1):
    import full_model  # noqa: F401  (the Model class)
    self.model, self.opt, self.d = model, model.opt, model.dims
    d = self.d
    if not torch.cuda.is_available():
      raise rn.RecAttendError('the training step needs an MI355X (HIP device); there is no CPU fallback')
    if self.opt.get('box_loss_fn', 'iou') not in ('iou', 'mse', 'huber') or \
        self.opt.get('segm_loss_fn', 'iou') not in ('iou', 'wt_cov'):
      # the reference's other branches cannot run: box 'wt_cov' feeds a scalar to f_weighted_coverage
      # (full_model.py:949,968), box 'bce' assigns to box_loss_fn (:971), segm 'bce' calls an undefined name (:1016)
      raise NotImplementedError('box_loss_fn in (iou, mse, huber) and segm_loss_fn in (iou, wt_cov) are the '
                                'branches that execute in the reference')
    self._setup(model, world)
    cmap_c, _ = model.engine._chan_map(d['ctrl_in'])
    cmap_a, _ = model.engine._chan_map(d['attn_in'])
    self.cmap_c = None if cmap_c == list(range(len(cmap_c))) else cmap_c
    self.cmap_a = None if cmap_a == list(range(len(cmap_a))) else cmap_a

  def _setup(self, model, world):
    """Bucket, autograd leaves, flat BN statistics, per-trainer caches; then every rank takes rank 0's state."""
    import torch.distributed as dist
    self.bucket = GradBucket(model)
    self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else int(world)
    self._pack = _PackCache(self)  # this trainer's per-step cache of packed filters / padded biases / packed LSTM weights
    self._wgrad_parts = _DeferredWgrads()
    self._slabs, self._bn_tabs = {}, {}  # the batched-backward step's [T, ...] buffers and per-layer BatchNorm pointer tables
    self._graphs = {}
    self.leaves = {}
    for k in self.bucket.names:
      leaf = model[k].detach().requires_grad_(True)  # shares the bucket's storage
      leaf.grad = self.bucket.grad_of[k]            # autograd accumulates straight into the bucket
      self.leaves[k] = leaf
    _flat_bn_statistics(self)
    # whole-batch BatchNorm statistics under data parallelism (nnlib.py:98): per BN call one all_gather forward and
    # one all-reduce backward.  Collectives inside a captured HIP graph have not been exercised on this stack, so
    # the synchronised step runs eagerly.
    self.sync_bn = bool(self.opt.get('sync_bn', False)) and self.world > 1
    # model_opt['compute_dtype']: 'float32' (the reference's arithmetic) or 'bf16' — the conv layers' forward, data
    # and filter gradients take bf16 operands on the bf16 MFMA with float32 accumulation; master weights, Adam state,
    # activations, BatchNorm, the loss and every other kernel stay float32
    cd = str(self.opt.get('compute_dtype', 'float32')).lower()
    if cd not in ('float32', 'f32', 'fp32', 'bf16', 'bfloat16'):
      raise rn.RecAttendError("model_opt['compute_dtype'] = %r: 'float32' or 'bf16'" % cd)
    self.bf16 = cd in ('bf16', 'bfloat16')
    # ... and, in the stacked step, U / Y / dY / dU of the conv layers STORED as bf16 between their passes (half the bytes of
    # the HBM-bound BatchNorm, data-gradient and filter-gradient passes); RA_BF16_STORE=0: operands only, float32 tensors
    self.bf16_store = False
    if self.sync_bn:
      self.use_graph = False
    rank = dist.get_rank() if self.world > 1 else 0
    self.aug_gen = torch.Generator().manual_seed(int(self.opt.get('seed', 1234)) + 7919 * rank)  # augmentation draws (CPU)
    self.broadcast_state()

  def broadcast_state(self, src=0):
    """Rank `src`'s weights, Adam moments, global_step and BN EMA shadows to every rank (the ranks' own
    initial draws differ)."""
    import torch.distributed as dist
    if self.bucket.broadcast(src) > 1:
      dist.broadcast(self.ema, src=src)
      eng = getattr(self.model, 'engine', None)
      if eng is not None:
        eng._stamp = None

  # ------------------------------------------------------------------ checkpoint (utils/saver.py:24-31, experiment.py:26-37)
  def state_dict(self):
    """Everything a restart needs beyond Model.state_dict_numpy() (weights + EMA shadows): Adam m / v per parameter
    and global_step (learn-rate staircase, knob schedules)."""
    self.flush_status()  # a checkpoint is not written over a step whose matching failed
    return self.bucket.state_dict()

  def load_state_dict(self, state, strict=True):
    self.bucket.load_state_dict(state, strict=strict)
    self._graphs = {}  # a captured step froze nothing of this, but start clean

  # ------------------------------------------------------------------ pieces
  def _grad_views(self, scope, i, key, bn):
    """The layer's parameter gradients go straight into the bucket (ConvBNActPool.backward, in-kernel)."""
    if not (self.fuse_param_grads and bn and torch.is_grad_enabled()):
      return None
    g = self.bucket.grad_of
    names = ('%s_w_%d' % (scope, i), '%s_b_%d' % (scope, i), key + '_gamma', key + '_beta')
    if not all(n in g for n in names):
      return None
    return tuple(g[n] for n in names)

  fuse_param_grads = True
  defer_wgrad = True  # one finishing reduction of the filter gradient per layer and step (not one per timestep)

  match_side_stream = os.environ.get('RA_MATCH_SIDE', '1') != '0'  # the box matching under the mask matching (one fork / join)
  match_merged = os.environ.get('RA_MATCH_MERGED', '1') != '0'  # otherwise: both matchings as one launch of 2 B problems

  def _cnn(self, x, scope, n, pools, tt, cmap0, stats):
    P, hs = self.leaves, []
    for i in range(n):
      bn = self.d['use_bn']
      key = '%s_%d_%d' % (scope, i, tt)
      meta = dict(transposed=False, stride=1, pool=pools[i], relu=True, chan_map=cmap0 if i == 0 else None,
                  bf16_store=self.bf16_store, y_f32=(i == n - 1),  # the net's last output feeds float32 kernels (controller, score, dcnn)
                  stat_out=self._stat_views.get(key), grads=self._grad_views(scope, i, key, bn), cache=self._pack,
                  sync_bn=self.sync_bn, bf16=self.bf16, wgrad_defer=self._wgrad_parts if self.defer_wgrad else None,
                  wgrad_key=(scope, i), alloc=self._tape_alloc(scope, i, tt, meta_of=(False, 1, pools[i], cmap0 if i == 0 else None)))
      x, mean, var = ConvBNActPool.apply(_pad_channels(x) if i else x, P['%s_w_%d' % (scope, i)], P['%s_b_%d' % (scope, i)],
                                         P[key + '_gamma'] if bn else None, P[key + '_beta'] if bn else None, meta)
      if bn:
        stats[key] = (mean, var)
      hs.append(x)
    return hs

  def _dcnn(self, x, scope, n, unpool, tt, stats, skips=None):
    """nnlib.dcnn (nnlib.py:362-400).  skips[i] = (tensor, real channel map or None) is concatenated
    behind the previous layer's output (concat(prev, skip), :365): one packed kernel input whose
    chan_map sends every packed channel to its row of the [3,3,out,in] filter."""
    P = self.leaves
    for i in range(n):
      bn = self.d['use_bn']
      key = '%s_%d_%d' % (scope, i, tt)
      cmap = None
      if skips is not None and skips[i] is not None:
        sk, smap = skips[i]
        prev_c = x.shape[3]
        xp, skp = _pad_channels(x), _pad_channels(sk if sk.dtype == x.dtype else sk.to(x.dtype))
        smap = list(range(sk.shape[3])) if smap is None else smap
        cmap = list(range(prev_c)) + [-1] * (xp.shape[3] - prev_c) + [prev_c + m if m >= 0 else -1 for m in smap] + \
            [-1] * (skp.shape[3] - len(smap))
        x = torch.cat([xp, skp], dim=3)
      meta = dict(transposed=True, stride=unpool[i], pool=1, relu=True, chan_map=cmap, stat_out=self._stat_views.get(key),
                  bf16_store=self.bf16_store, y_f32=(i == n - 1),
                  grads=self._grad_views(scope, i, key, bn), cache=self._pack, sync_bn=self.sync_bn, bf16=self.bf16,
                  wgrad_defer=self._wgrad_parts if self.defer_wgrad else None, wgrad_key=(scope, i),
                  alloc=self._tape_alloc(scope, i, tt, meta_of=(True, unpool[i], 1, cmap)))
      x, mean, var = ConvBNActPool.apply(_pad_channels(x), P['%s_w_%d' % (scope, i)], P['%s_b_%d' % (scope, i)],
                                         P[key + '_gamma'] if bn else None, P[key + '_beta'] if bn else None, meta)
      if bn:
        stats[key] = (mean, var)
    return x

  def _linear(self, x, wname, bname):
    """x W + b for two parameters of the bucket; their gradients accumulate in place when the step fuses them."""
    P, g = self.leaves, self.bucket.grad_of
    if self.fuse_param_grads and torch.is_grad_enabled() and wname in g and bname in g:
      return LinearAcc.apply(x, P[wname], P[bname], g[wname], g[bname])
    return torch.addmm(P[bname], x, P[wname])

  # ------------------------------------------------------------------ the T timesteps' backward passes, stacked
  # full_model's timesteps are coupled only through the canvas (gradient stopped, full_model.py:843-848; the controller
  # state starts from zero in each, :668-689).  So the step runs in two phases: (1) the sequential forward WITHOUT an
  # autograd tape, every conv layer writing its u / y into [T, ...] slabs; (2) the differentiable graph built ONCE over
  # T * B stacked images — the conv layers as ConvStackFn nodes over the slabs (no recomputation), the cheap per-image
  # pieces (controller, attention head, knob, extract, paste, score) re-run stacked — whose backward pass is then one
  # launch group per layer instead of one per layer and timestep.  Same numbers as the per-timestep graph up to
  # summation order; tests/test_train_gpu.py compares both with the float64 oracle.
  batched_backward = os.environ.get('RA_BATCHED_BWD', '1') != '0'
  _tape = None

  def _batched_ok(self, extra):
    """The stacked step covers every architecture of the run scripts — skip connections, d_in / y_in, use_iou_box, the
    'mse' / 'huber' box losses, --sync_bn — as long as the layers' channel counts are multiples of 4 (the slabs are the
    kernels' packed tensors) and the controller fits the fused kernels (ra_ctrl_train.hip: up to 4 layers per MLP)."""
    d, opt = self.d, self.opt
    c4 = lambda cs: all(c % 4 == 0 for c in cs)
    return bool(self.batched_backward and torch.is_grad_enabled() and d['use_bn'] and self.fuse_param_grads and
                self.fuse_controller and bool(opt.get('stop_canvas_grad', True)) and  # a canvas gradient couples the timesteps
                c4(self.model.dims['ccnn_channels'][1:]) and c4(opt['attn_cnn_depth']) and c4(opt['attn_dcnn_depth'][:-1]) and
                self.model.dims['C0p'] % 4 == 0 and
                rn.lib().ra_ctrl_train_supported_n(d['G'], self.model.dims['ccnn_channels'][-1], d['hid'], d['iters'], 9, int(d['n_gmlp']),
                                                   int(d['n_cmlp']), int(d.get('mlp_dim', 0) or 0)))

  def _slab(self, name, T, shape, dtype=torch.float32):
    """[T, *shape] buffer of the step (one slot per timestep), kept across steps."""
    t = self._slabs.get(name)
    if t is None or tuple(t.shape) != (T,) + tuple(shape) or t.dtype != dtype:
      if t is not None:
        self._drop_captured_steps()  # a graph captured for another batch shape holds the old slab's address
      t = self._slabs[name] = torch.empty((T,) + tuple(shape), dtype=dtype, device=self.bucket.param.device)
    return t

  def _drop_captured_steps(self):
    """Step-persistent scratch (the [T, ...] slabs, the fused controller's buffers) is shared by every batch shape and
    its addresses are baked into a captured step: when a new shape makes it move, the steps captured for other shapes
    must not be replayed again (they would read and write freed memory) — they are re-captured when their shape returns."""
    for key in [k for k, st in self._graphs.items() if 'graph' in st]:
      del self._graphs[key]

  def _tape_alloc(self, scope, i, tt, meta_of):
    """ConvBNActPool's output placement while the sequential phase runs: u / y of layer (scope, i) at timestep tt."""
    tape = self._tape
    if tape is None:
      return None
    tape['layers'][(scope, i)] = meta_of
    T = self.d['T']
    return lambda kind, shape, dtype=torch.float32: self._slab('%s_%d_%s' % (scope, i, kind), T, shape, dtype)[tt]

  def _bn_tables(self, scope, i):
    """Device table of 6 T pointers {mean, var, gamma, beta, gradient gamma, gradient beta}[T] of layer (scope, i) (static:
    the flat statistics buffer and the parameter / gradient bucket never move) + the same as per-timestep tensors."""
    hit = self._bn_tabs.get((scope, i))
    if hit is None:
      T, P, g = self.d['T'], self.leaves, self.bucket.grad_of
      keys = ['%s_%d_%d' % (scope, i, tt) for tt in range(T)]
      cols = [[self._stat_views[k][0] for k in keys], [self._stat_views[k][1] for k in keys], [P[k + '_gamma'] for k in keys],
              [P[k + '_beta'] for k in keys], [g[k + '_gamma'] for k in keys], [g[k + '_beta'] for k in keys]]
      tab = torch.tensor([t.data_ptr() for col in cols for t in col], dtype=torch.int64).to(self.bucket.param.device)
      hit = self._bn_tabs[(scope, i)] = (tab, list(zip(*cols)))
    return hit

  def _stack_layers(self, X, scope, n, skips=None):
    """Phase 2: the n layers of a net as ConvStackFn nodes over the slabs the sequential phase filled; returns every
    layer's output.  skips[i] (dcnn, nnlib.py:365): the tensor concatenated behind layer i's input, as _dcnn packs it."""
    T, P = self.d['T'], self.leaves
    hs = []
    for i in range(n):
      tr, stride, pool, cmap = self._tape['layers'][(scope, i)]
      if skips is not None and skips[i] is not None:
        sk = skips[i] if skips[i].dtype == X.dtype else skips[i].to(X.dtype)
        X = torch.cat([_pad_channels(X), _pad_channels(sk)], dim=3)  # the channel map of the sequential phase applies
      U, Y = self._slabs['%s_%d_u' % (scope, i)], self._slabs['%s_%d_y' % (scope, i)]
      tab, per_group = self._bn_tables(scope, i)
      gw, gb = self.bucket.grad_of['%s_w_%d' % (scope, i)], self.bucket.grad_of['%s_b_%d' % (scope, i)]
      info = dict(G=T, B=U.shape[1], U=U.view((-1,) + U.shape[2:]), Y=Y.view((-1,) + Y.shape[2:]), transposed=tr, stride=stride,
                  pool=pool, relu=True, chan_map=cmap, bf16=self.bf16, tabs=tab, per_group=per_group, gw=gw, gb=gb, cache=self._pack,
                  sync_world=self.world if self.sync_bn else 1)
      X = ConvStackFn.apply(X, P['%s_w_%d' % (scope, i)], P['%s_b_%d' % (scope, i)], info)
      hs.append(X)
    return hs

  def _stacked_graph(self, inp_slab, matches, knob_box, gt_windows, head_flags, cc, gt_corners=None):
    """Phase 2: the differentiable graph of all T timesteps at once, images stacked (t, b).  Returns y_out [B,T,H,W],
    s_out [B,T], attn_box [B,T,H,W] with their autograd history."""
    d, P = self.d, self.leaves
    T, H, W, Fh, Fw = d['T'], d['H'], d['W'], d['Fh'], d['Fw']
    B = inp_slab.shape[1]
    N = T * B
    inp_all = inp_slab.view((N,) + inp_slab.shape[2:])
    feat = self._stack_layers(inp_all, 'ctrl_cnn', d['ccnn_nlayers'])[-1]
    bufs = self._ctrl_buffers(B, feat.shape[3])
    Wg, bg = self._lstm_weights()
    h, co = ControllerFn.apply(feat.reshape(N, d['G'], -1), Wg.detach(), bg.detach(), self._mlp_weights('glimpse_mlp', d['n_gmlp']),
                               self._mlp_weights('ctrl_mlp', d['n_cmlp']), bufs, 'all')
    cn, ls, ctr, size, lg_var, ag, bgm, ylg, arec = AttnHead.apply(co, H, W, Fh, Fw, head_flags)
    # the planes the sequential phase computed (same window parameters, timestep by timestep): the nodes only add the backward
    pre = (lambda name, shape: [self._slab(name, T, shape).view((N,) + shape[1:])]) if self.reuse_attn_planes else (lambda name, shape: None)
    box = AttnPaste.apply(None, ctr, size, lg_var, bgm, H, W, Fh, Fw, None, pre('box', (B, H, W)), [arec])
    iou_rows = None
    if gt_corners is not None:  # use_knob + use_iou_box: the [B,T,T] matrix of corner IoUs the boxes are matched and scored on
      import modellib          # (modellib.f_iou_box, full_model.py:750-754,931-934), differentiable through the predicted corners
      gc = gt_corners.unsqueeze(0).expand((T,) + tuple(gt_corners.shape)).reshape((N,) + tuple(gt_corners.shape[1:]))
      iou_rows = modellib.f_iou_box((ctr - size / 2.0)[:, None], (ctr + size / 2.0)[:, None], gc[:, :, 0:2], gc[:, :, 2:4])
      iou_rows = iou_rows.view(T, B, T).transpose(0, 1).contiguous()
    if knob_box is not None:  # the matches of phase 1 are constants of the graph (they were never differentiated)
      match_all = torch.stack(matches, dim=0).reshape(N, T)
      rep = lambda t: t.unsqueeze(0).expand((T,) + tuple(t.shape)).reshape((N,) + tuple(t.shape[1:])).contiguous()
      knob_all = knob_box.reshape(B, T).t().reshape(N).contiguous()
      ctr, size, arec = KnobMix.apply(ctr, size, match_all, rep(gt_windows[0]), rep(gt_windows[1]), knob_all, arec)
    x_patch = AttnExtract.apply(inp_all, ctr, size, lg_var, ag, Fh, Fw, None, pre('xpatch', (B, Fh, Fw, inp_all.shape[3])), [arec])
    h_acnn = self._stack_layers(x_patch, 'attn_cnn', d['acnn_nlayers'])
    core = h_acnn[-1]
    skips = None
    if d['skip_ch'] is not None and any(d['skip_ch']):   # full_model.py:798-805: reversed CNN outputs, then x_patch
      rev = h_acnn[::-1][1:] + [x_patch]
      skips = [None] + [rev[i - 1] if (i - 1 < len(rev) and d['skip_ch'][i]) else None for i in range(1, d['adcnn_nlayers'])]
    y_patch = self._stack_layers(core, 'attn_dcnn', d['adcnn_nlayers'], skips)[-1]
    y = AttnPaste.apply(y_patch if y_patch.shape[-1] == 1 else y_patch[..., 0:1], ctr, size, lg_var, ylg, H, W, Fh, Fw, None, pre('ymask', (B, H, W)), [arec])
    if d['disable_overwrite']:
      y = (1.0 - inp_all[..., cc]) * y
    s = torch.sigmoid(self._linear(torch.cat([h, core.reshape(N, -1)], dim=1), 'score_mlp_w_0', 'score_mlp_b_0'))
    to_bt = lambda t: t.view((T, B) + tuple(t.shape[1:])).transpose(0, 1).contiguous()
    # the masks and boxes stay timestep-major [T,B,H,W]: the pairwise IoU reads them (and writes their gradient) through strides
    return y.view(T, B, H, W), to_bt(s).reshape(B, T), box.view(T, B, H, W), to_bt(cn), to_bt(ls), iou_rows

  fuse_controller = True  # the controller of a timestep as one forward and one backward launch
  fused_loss_head = os.environ.get('RA_LOSS_HEAD', '1') != '0'  # the scalar loss head as one launch each way (LossHead); False: the autograd graph of element-wise ops
  reuse_attn_planes = True  # stacked step: box / patch / mask planes of the sequential phase feed the stacked graph (no second forward)
  bf16_storage = os.environ.get('RA_BF16_STORE', '1') != '0'  # bf16 mode, stacked step: U / Y / dY / dU stored as bf16 (False: operands only)
  seq_ctrl_split = os.environ.get('RA_TRAIN_CTRL_SPLIT', '1') != '0'  # stacked step, sequential phase: the decode loop's 16-workgroup controller

  def _seq_controller(self, B):
    """The sequential phase of the stacked step keeps nothing of the controller but its outputs (the stacked graph re-runs
    it over all timesteps with the state the backward needs), so it can run the DECODE loop's 16-workgroup controller
    (csrc/ra_ctrl_split.hip: 63 us per launch against 136 for ra_ctrl_train_fwd_f32's one workgroup per image).  Its
    weights are packed on the device once per optimisation step: the host packer only moves values, so packing arrays
    of flat-bucket POSITIONS once gives an index map, and a step's packing is one gather launch.  None where the split
    form does not apply (descriptor, more than 14 images)."""
    if not self.seq_ctrl_split:
      return None
    sc = getattr(self, '_seqc', None)
    if sc is None or sc.get('B') != B:
      d, dev = self.d, self.bucket.param.device
      Cf = self.model.dims['ccnn_channels'][-1]
      desc = ops.make_ctrl_desc(d['G'], Cf, d['hid'], d['iters'], d['n_gmlp'], d['n_cmlp'], d['mlp_dim'], d['H'], d['W'], d['Fh'],
                                d['Fw'], d['squash'], d['fixed_var'], d['dynamic_var'], d.get('fixed_gamma', True))
      sc = self._seqc = {'B': B, 'ok': False}
      if ops.ctrl_split_supported(desc) and B * 16 <= ops.cu_count() - 32:
        off = self.bucket.offsets
        pos = lambda k: (off[k][0] + 1 + np.arange(off[k][1], dtype=np.float64)).astype(np.float32).reshape(off[k][2])
        assert self.bucket.param.numel() < (1 << 24)  # positions are exact in float32
        lstm = {k: pos('ctrl_lstm_' + k) for k in ('w_xi', 'w_hi', 'b_i', 'w_xf', 'w_hf', 'b_f', 'w_xu', 'w_hu', 'b_u', 'w_xo', 'w_ho', 'b_o')}
        gm = [(pos('glimpse_mlp_w_%d' % i), pos('glimpse_mlp_b_%d' % i)) for i in range(d['n_gmlp'])]
        cm = [(pos('ctrl_mlp_w_%d' % i), pos('ctrl_mlp_b_%d' % i)) for i in range(d['n_cmlp'])]
        imap = ops.pack_ctrl_split_weights(desc, lstm, gm, cm).astype(np.int64) - 1
        ws, status = ops.ctrl_split_workspace(desc, B, dev)
        f = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)
        sc.update(ok=True, desc=desc, imap=torch.as_tensor(imap.astype(np.int32)).to(dev), wp=f(imap.size), ws=ws, status=status,
                  h=f(d['T'], B, d['hid']), co=f(d['T'], B, 9), gm=f(B, d['iters'], d['G']), attn=f(B, rn.RA_ATTN_STRIDE), packed_at=None)
    if not sc['ok']:
      return None
    if sc['packed_at'] != self._pack_epoch:  # once per optimisation step (forward_loss bumps the epoch)
      check(rn.lib().ra_gather_f32(ptr(self.bucket.param), ptr(sc['imap']), sc['imap'].numel(), ptr(sc['wp']), rn.stream_ptr()), 'ra_gather_f32')
      sc['packed_at'] = self._pack_epoch
    return sc

  def _ctrl_buffers(self, B, Cf):
    """The fused controller's step buffers, or None where the library path has to run: shapes beyond the kernel's LDS or
    its four layers per MLP, no gradient bucket views."""
    d, g = self.d, self.bucket.grad_of
    names = ['glimpse_mlp_%s_%d' % (p, l) for l in range(d['n_gmlp']) for p in 'wb'] + \
        ['ctrl_mlp_%s_%d' % (p, l) for l in range(d['n_cmlp']) for p in 'wb'] + \
        ['ctrl_lstm_%s%s' % (p, k) for p in ('w_x', 'w_h', 'b_') for k in 'ifou']
    if not (self.fuse_controller and self.fuse_param_grads and all(n in g for n in names) and
            rn.lib().ra_ctrl_train_supported_n(d['G'], Cf, d['hid'], d['iters'], 9, int(d['n_gmlp']), int(d['n_cmlp']),
                                               int(d.get('mlp_dim', 0) or 0))):
      return None
    cb = getattr(self, '_ctl', None)
    if cb is None or cb.B != B or cb.T != d['T']:
      if cb is not None:
        self._drop_captured_steps()
      cb = self._ctl = _CtrlStepBuffers(self, d['T'], B)
    return cb

  def _mlp_weights(self, scope, n):
    """[(w, b), ...] of an MLP's n layers, detached (ControllerFn's plain-tensor arguments)."""
    P = self.leaves
    return [(P['%s_w_%d' % (scope, l)].detach(), P['%s_b_%d' % (scope, l)].detach()) for l in range(n)]

  def _controller(self, feat, tt=None):
    """full_model.py:668-689 on [B,G,Cf] features: glimpse read-out, LSTM (state = [c|h], zeroed per
    timestep), glimpse MLP (softmax over G), controller MLP."""
    P, d = self.leaves, self.d
    B, G, hid = feat.shape[0], d['G'], d['hid']
    bufs = self._ctrl_buffers(B, feat.shape[2]) if tt is not None else None
    if bufs is not None:  # one launch forward, one backward (csrc/ra_ctrl_train.hip)
      Wg, bg = self._lstm_weights()
      return ControllerFn.apply(feat, Wg.detach(), bg.detach(), self._mlp_weights('glimpse_mlp', d['n_gmlp']),
                                self._mlp_weights('ctrl_mlp', d['n_cmlp']), bufs, tt)
    dev = feat.device  # constants of the recurrence's start: built once, never written
    c = h = _const('zeros', (B, hid), dev, lambda: torch.zeros((B, hid), device=dev))
    gmap = _const('gmap0', (B, G), dev, lambda: torch.full((B, 1, G), 1.0 / G, device=dev))
    Wg, bg = self._lstm_weights()
    acc = self._lstm_grad_acc(Wg, bg)
    for it in range(d['iters']):
      glimpse = torch.bmm(gmap, feat).reshape(B, -1)              # sum_g map[g] feat[g, :]  (a view: `[:, 0]` costs a zero-fill + copy backward)
      xh = torch.cat([glimpse, h], dim=1)
      pre = LinearAcc.apply(xh, Wg, bg, *acc) if acc is not None else torch.addmm(bg, xh, Wg)   # all four gates: one GEMM
      h, c = LSTMCell.apply(pre, c)
      if it < d['iters'] - 1:
        z = h
        for l in range(d['n_gmlp']):
          z = self._linear(z, 'glimpse_mlp_w_%d' % l, 'glimpse_mlp_b_%d' % l)
          z = torch.relu(z) if l < d['n_gmlp'] - 1 else torch.softmax(z, dim=1)
        gmap = z[:, None, :]
    z = h
    for l in range(d['n_cmlp']):
      z = self._linear(z, 'ctrl_mlp_w_%d' % l, 'ctrl_mlp_b_%d' % l)
      if l < d['n_cmlp'] - 1:
        z = torch.relu(z)
    return h, z

  def _lstm_grad_acc(self, Wg, bg):
    """(gW, gb, end-of-backward scatter) for the packed gate weights: step-local gradient buffers, zeroed once per
    step, whose blocks are added to the eight parameter gradients when the backward pass ends.  None: plain autograd."""
    g = self.bucket.grad_of
    names = ['ctrl_lstm_w_x' + k for k in 'ifou'] + ['ctrl_lstm_w_h' + k for k in 'ifou'] + ['ctrl_lstm_b_' + k for k in 'ifou']
    if not (self.fuse_param_grads and torch.is_grad_enabled() and all(n in g for n in names)):
      return None
    hit = self._pack.get('lstm_acc')
    if hit is None:
      buf = getattr(self, '_lstm_gbuf', None)
      if buf is None or buf[0].shape != Wg.shape:
        buf = self._lstm_gbuf = (torch.empty_like(Wg), torch.empty_like(bg))
      buf[0].zero_()
      buf[1].zero_()
      hit = self._pack['lstm_acc'] = (buf[0], buf[1], _LstmGradScatter(self, buf[0], buf[1]))
    return hit

  def _lstm_weights(self):
    """[w_x ; w_h] of the four gates side by side (i, f, o, u) and their biases: built once per step (the
    weights are shared by all timesteps and glimpses), so a cell is one GEMM + one pointwise kernel."""
    hit = self._pack.get('lstm')
    if hit is None:
      P = self.leaves
      Wg = torch.cat([torch.cat([P['ctrl_lstm_w_x' + g], P['ctrl_lstm_w_h' + g]], dim=0) for g in 'ifou'], dim=1)
      bg = torch.cat([P['ctrl_lstm_b_' + g] for g in 'ifou'], dim=0)
      hit = self._pack['lstm'] = (Wg, bg)
    return hit

  # ------------------------------------------------------------------ forward + loss
  def draw_knobs(self, B, generator=None, out=None):
    """The random draws of one training step (full_model.py:567-577,612-625,829-831): GT-box padding
    and centre noise, the two Bernoulli knobs, the per-timestep segmentation noise.  One generator
    per rank (seeded rank-offset by the caller) keeps data-parallel ranks decorrelated.
    out: the captured step's static buffers — the [T,B,H,W] noise plane is drawn straight into its buffer
    (uniform_(0, a) = a * rand bit for bit, same generator consumption: one launch instead of draw + scale + a 134 MB copy
    at cfg4), the B T-sized draws are copied."""
    d, opt = self.d, self.opt
    dev = self.bucket.param.device
    T, H, W = d['T'], d['H'], d['W']
    u = lambda *s: torch.rand(s, generator=generator, device=dev)
    pr, pn, cn = float(opt['attn_box_padding_ratio']), float(opt['gt_box_pad_noise']), float(opt['gt_box_ctr_noise'])
    small = {'pad': pr - pn + 2 * pn * u(B, T, 1), 'shift': -cn + 2 * cn * u(B, T, 2), 'u_box': u(B, T, 1), 'u_segm': u(B, T, 1)}
    if out is None:
      small['segm_noise'] = float(opt['gt_segm_noise']) * u(T, B, H, W)
      return small
    for k, v in small.items():
      out[k].copy_(v)
    out['segm_noise'].uniform_(0.0, float(opt['gt_segm_noise']), generator=generator)
    return out

  def knob_shapes(self, B):
    d = self.d
    return {'pad': (B, d['T'], 1), 'shift': (B, d['T'], 2), 'u_box': (B, d['T'], 1), 'u_segm': (B, d['T'], 1),
            'segm_noise': (d['T'], B, d['H'], d['W'])}

  fused_knob_setup = os.environ.get('RA_KNOB_SETUP', '1') != '0'  # one launch on gt_box's partials (ra_knob_setup_f32)

  def _knob_setup(self, y_gt, knobs, gt_ws=None):
    """Noisy GT attention (modellib.get_gt_attn with tensor padding / centre shift,
    full_model.py:567-577) and the knob masks (:596-625) at the current global step.  gt_ws: the workspace of the
    step's ops.gt_box call (same y_gt, min_padding = padding + 4): the fused form reads its min / max / sum partials
    instead of reducing y_gt twice more and chaining ~28 element-wise launches."""
    d, opt = self.d, self.opt
    T = d['T']
    dev = y_gt.device
    mp = float(opt['padding']) + 4.0
    if gt_ws is not None and self.fused_knob_setup:
      sched = getattr(self, '_sched', None)
      if sched is None:
        step = self.bucket.global_step
        sched = torch.tensor([knob_prob(opt, step, opt['knob_box_offset']), knob_prob(opt, step, opt['knob_segm_offset'])],
                             dtype=torch.float32, device=dev)
      return ops.knob_setup(gt_ws, y_gt.shape[0], T, knobs['pad'], knobs['shift'], knobs['u_box'], knobs['u_segm'], sched, mp,
                            opt.get('knob_use_timescale', False))
    raw, _ = ops.gt_box(y_gt, 0.0, 0.0, want_box=False)          # raw min / max indices (0 for empty instances)
    tl, br = raw[:, :, 0:2], raw[:, :, 2:4]
    nz = (ops.pair_stats(y_gt, y_gt, want=('sum_b',))['sum_b'] > 0).to(torch.float32)[:, :, None]
    size = br - tl
    padv = torch.clamp(knobs['pad'] * size, min=mp)              # modellib.py:688-691
    tl_n = (tl + knobs['shift'] * size - padv) * nz
    br_n = nz * (br + knobs['shift'] * size + padv) + (1 - nz) * (2 * mp)
    ctr_n, size_n = (tl_n + br_n) / 2.0, br_n - tl_n
    if opt.get('knob_use_timescale', False):
      scale = 1.0 + torch.log(1.0 + torch.arange(T, dtype=torch.float32, device=dev) * 3.0)
    else:
      scale = torch.ones(T, device=dev)
    step = self.bucket.global_step
    sched = getattr(self, '_sched', None)  # graph replay: the two probabilities live in a device tensor
    kb0 = sched[0] if sched is not None else knob_prob(opt, step, opt['knob_box_offset'])
    ks0 = sched[1] if sched is not None else knob_prob(opt, step, opt['knob_segm_offset'])
    pb = torch.clamp(kb0 * scale, max=1.0)[None, :, None]
    ps = torch.clamp(ks0 * scale, max=1.0)[None, :, None]
    return ctr_n, size_n, (knobs['u_box'] <= pb).to(torch.float32), (knobs['u_segm'] <= ps).to(torch.float32)

  def forward_loss(self, x, y_gt, s_gt, knobs=None, generator=None, d_in=None, y_in=None):
    """The training graph (phase_train = True): returns (total loss, dict of pieces, BN batch stats).
    With model_opt use_knob the ground truth is mixed in (full_model.py:744-773,826-841) using the
    draws in `knobs` (draw_knobs() when None).  d_in / y_in: the extra input channels of the KITTI /
    Cityscapes architectures (full_model.py:165-194)."""
    P, d, opt = self.leaves, self.d, self.opt
    self._pack.clear()  # the optimizer wrote new weights since the last step
    self._match_seq = 0  # (_segm_match's staging blocks are keyed by the call's position in the step)
    self._pack_epoch = getattr(self, '_pack_epoch', 0) + 1
    self._pack.refresh()  # this step's packed filters and padded biases: one gather over the parameter bucket
    self._wgrad_parts.reset()
    if getattr(self, '_ctl', None) is not None:
      self._ctl.begin_step()
    dev = self.bucket.param.device
    as_t = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float32))).to(
        device=dev, dtype=torch.float32).contiguous()
    x, y_gt, s_gt = as_t(x), as_t(y_gt), as_t(s_gt)
    extra = []
    if d['add_d_out']:
      if d_in is None or y_in is None:
        raise rn.RecAttendError('this architecture feeds d_in and y_in (full_model.py:165-194)')
      extra = [as_t(d_in), as_t(y_in)]
    B, T, H, W, Fh, Fw = x.shape[0], d['T'], d['H'], d['W'], d['Fh'], d['Fw']
    use_knob = bool(opt.get('use_knob', False))
    fixed = bool(opt.get('fixed_order', False))
    gt_corners, box_gt, gt_ws = ops.gt_box(y_gt, float(opt['attn_box_padding_ratio']), float(opt['padding']) + 4.0, want_ws=True)
    if use_knob:
      if knobs is None:
        knobs = self.draw_knobs(B, generator)
      ctr_gtn, size_gtn, knob_box, knob_segm = self._knob_setup(y_gt, knobs, gt_ws)
    canvas = torch.zeros((B, H, W, 1), device=dev)
    stats, y_list, s_list, box_list, cn_list, ls_list, iou_box_steps = {}, [], [], [], [], [], []
    dims_hw = _const('dims', (H, W), dev, lambda: torch.tensor([H, W], dtype=torch.float32, device=dev))
    head_flags = (1 if d['squash'] else 0) | (2 if d['fixed_var'] else 0) | (4 if d['dynamic_var'] else 0) | (8 if d['fixed_gamma'] else 0)
    cc = x.shape[3]  # the canvas channel of the packed input
    canvas_grad = not bool(opt.get('stop_canvas_grad', True)) and torch.is_grad_enabled()
    inp = torch.cat([x, canvas] + extra, dim=3)   # packed [x | canvas | d_in | y_in], zero-padded to C0p
    if inp.shape[3] != d['C0p']:
      inp = _pad_channels(inp)
    batched = self._batched_ok(extra) and self._ctrl_buffers(B, self.model.dims['ccnn_channels'][-1]) is not None
    self._tape = dict(layers={}) if batched else None
    self.bf16_store = bool(batched and self.bf16 and not self.sync_bn and self.bf16_storage)
    tape_match = []
    if batched:  # phase 1 runs without an autograd tape; the packed inputs of the T timesteps live in one slab
      inp_slab = self._slab('inp', T, tuple(inp.shape))
      inp_slab[0].copy_(inp)
      inp = inp_slab[0]
    with (torch.no_grad() if batched else contextlib.nullcontext()):
      for tt in range(T):
        feat = self._cnn(inp, 'ctrl_cnn', d['ccnn_nlayers'], d['ccnn_pool'], tt, self.cmap_c, stats)[-1]
        sc = self._seq_controller(B) if batched else None
        if sc is not None:  # outputs only: the decode loop's controller (its own h / ctrl_out rows per timestep)
          h, co = sc['h'][tt], sc['co'][tt]
          ops.controller_split(sc['desc'], feat.reshape(B, d['G'], -1), sc['wp'], h, co, sc['gm'], sc['attn'], sc['ws'], sc['status'])
        else:
          h, co = self._controller(feat.reshape(B, d['G'], -1), tt)
        # controller output -> window centre / size / variance and the three gammas (modellib.py:752-764,812-825): one launch
        cn, ls, ctr, size, lg_var, ag, bgm, ylg, arec = AttnHead.apply(co, H, W, Fh, Fw, head_flags)
        # attention box: sigmoid(box_gamma * extract_patch(ones, F_y^T, F_x^T) - 5) (full_model.py:738-741)
        keep = batched and self.reuse_attn_planes  # the planes go straight into the [T, ...] slabs the stacked graph reads
        box = AttnPaste.apply(None, ctr, size, lg_var, bgm, H, W, Fh, Fw, self._slab('box', T, (B, H, W))[tt] if keep else None, None, [arec])
        if use_knob:  # kick in the (noisy) ground-truth box; lg_var keeps the PREDICTED size (:702-709 run earlier)
          if fixed:
            gmatch = None
            gsel_box = _const('onehot', (B, T, tt), dev, lambda: torch.nn.functional.one_hot(
                torch.full((B,), tt, device=dev), T).to(torch.float32))
          else:
            if opt.get('use_iou_box', False):  # IoU of the box corners (modellib.f_iou_box, full_model.py:750-754)
              import modellib
              iou_row = modellib.f_iou_box((ctr - size / 2.0)[:, None], (ctr + size / 2.0)[:, None], gt_corners[:, :, 0:2],
                                           gt_corners[:, :, 2:4])
              iou_box_steps.append(iou_row[:, None, :])  # the row of the [B,T,T] matrix the box matching and loss use (:931-934)
              iou_t = iou_row.detach().contiguous()
            else:
              # f_iou of the box against the T ground-truth rectangles: one read of the box (not of the T planes too)
              iou_t = ops.box_iou_rects(box.detach(), gt_corners) if W % 4 == 0 and T <= 32 else \
                  ops.pair_stats(box.detach()[:, None].contiguous(), box_gt, want=('iou_soft',))['iou_soft']
            gmatch = ops.greedy_match(iou_t.view(B, T))              # matched set is never accumulated (:589,756)
            gsel_box = gmatch
          # (ctr, size) <- knob * matched noisy GT box + (1 - knob) * prediction: one launch (ra_knob_mix_f32)
          ctr, size, arec = KnobMix.apply(ctr, size, gsel_box, ctr_gtn, size_gtn, knob_box[:, tt], arec)
        # (batched: the attention CNN's first input of every timestep, contiguous over T for the stacked filter gradient)
        x_patch = AttnExtract.apply(inp if canvas_grad else inp.detach(), ctr, size, lg_var, ag, Fh, Fw,
                                    self._slab('xpatch', T, (B, Fh, Fw, inp.shape[3]))[tt] if batched else None, None, [arec])
        if batched and use_knob:
          tape_match.append(gsel_box)
        h_acnn = self._cnn(x_patch, 'attn_cnn', d['acnn_nlayers'], d['acnn_pool'], tt, self.cmap_a, stats)
        core = h_acnn[-1]
        skips = None
        if d['skip_ch'] is not None and any(d['skip_ch']):   # full_model.py:798-805: reversed CNN outputs, then x_patch
          full_a, _ = self.model.engine._chan_map(d['attn_in'])
          rev = [(hh, None) for hh in h_acnn[::-1][1:]] + [(x_patch, full_a)]
          skips = [None] + [rev[i - 1] if (i - 1 < len(rev) and d['skip_ch'][i]) else None
                            for i in range(1, d['adcnn_nlayers'])]
        y_patch = self._dcnn(core, 'attn_dcnn', d['adcnn_nlayers'], d['adcnn_unpool'], tt, stats, skips)
        y = AttnPaste.apply(y_patch if y_patch.shape[-1] == 1 else y_patch[..., 0:1], ctr, size, lg_var, ylg, H, W, Fh, Fw,
                            self._slab('ymask', T, (B, H, W))[tt] if keep else None, None, [arec])  # [B,H,W]
        if d['disable_overwrite']:
          y = (1.0 - inp[..., cc]) * y
        # (the score is not an input of the next timestep: the stacked graph computes it once for all timesteps)
        s = None if batched else torch.sigmoid(self._linear(torch.cat([h, core.reshape(B, -1)], dim=1), 'score_mlp_w_0', 'score_mlp_b_0'))
        # canvas <- max(y_c, canvas) with y_c = y, or with the knob: the (noisy) matched ground-truth segmentation mixed in
        # (:826-848, stop_canvas_grad) — and the next timestep's packed input, in one launch (ra_canvas_step_f32)
        if canvas_grad:
          # stop_canvas_grad = False (full_model.py:843-848): canvas = max(y_c, canvas) stays on the autograd tape — its
          # gradient reaches this timestep's mask (and, through earlier canvases, every mask before it) from the next
          # controller CNN's first layer, the next extract, and the (1 - canvas) factor of disable_overwrite
          y_c = y
          if use_knob:
            gsel = (gsel_box[:, :, None, None] * y_gt).sum(dim=1)
            gsel = gsel - gsel * knobs['segm_noise'][tt]
            ksv = knob_segm[:, tt].reshape(B, 1, 1)
            y_c = ksv * gsel + (1.0 - ksv) * y
          canvas_new = torch.maximum(y_c, inp[..., cc])
          inp = torch.cat([inp[..., :cc], canvas_new[..., None], inp[..., cc + 1:]], dim=3)
          y_list.append(y)
          s_list.append(s)
          box_list.append(box)
          cn_list.append(cn)
          ls_list.append(ls)
          continue
        nxt = inp_slab[tt + 1] if (batched and tt + 1 < T) else torch.empty_like(inp)
        yd = y.detach().contiguous()
        if use_knob:
          noise, ks = knobs['segm_noise'][tt].contiguous(), knob_segm[:, tt]
          check(rn.lib().ra_canvas_step_f32(ptr(inp), inp.shape[3], cc, B, H * W, ptr(yd), ptr(gsel_box), ptr(y_gt), T, ptr(noise),
                                            ptr(ks), int(ks.stride(0)), ptr(nxt), rn.stream_ptr()), 'ra_canvas_step_f32')
        else:
          check(rn.lib().ra_canvas_step_f32(ptr(inp), inp.shape[3], cc, B, H * W, ptr(yd), None, None, T, None, None, 1, ptr(nxt),
                                            rn.stream_ptr()), 'ra_canvas_step_f32')
        inp = nxt
        y_list.append(y)
        s_list.append(s)
        box_list.append(box)
        cn_list.append(cn)
        ls_list.append(ls)
    cn_bt = ls_bt = iou_rows_stacked = None
    if batched:
      y_out, s_out, attn_box, cn_bt, ls_bt, iou_rows_stacked = self._stacked_graph(
          inp_slab, tape_match, knob_box if use_knob else None, (ctr_gtn, size_gtn) if use_knob else None, head_flags, cc,
          gt_corners if (use_knob and not fixed and opt.get('use_iou_box', False)) else None)
      self._tape = None
    else:
      y_out, s_out = torch.stack(y_list, dim=1), torch.cat(s_list, dim=1)
      attn_box = torch.stack(box_list, dim=1)
    # ---- losses (full_model.py:913-1035), box_loss_fn = segm_loss_fn = 'iou'.  With the knob the
    # reference stacks the per-timestep box IoUs (:931-934): for use_iou_box = False (f_inter / f_union of the
    # predicted box against every GT box) the same numbers as the pairwise f_iou; with use_iou_box the stacked
    # corner IoUs (iou_box_steps, differentiable through the corners) are the matrix.
    tm = bool(batched)  # the stacked graph's masks / boxes are timestep-major [T,B,H,W] (PairIoU reads them through strides)
    ident = torch.eye(T, device=dev)[None] * s_gt[:, None, :] * s_gt[:, :, None]

    def matched_iou(a, b, iou=None, pre=None):
      iou = PairIoU.apply(a, b, tm) if iou is None else iou
      if fixed:
        m = ident
      elif pre is not None:  # matched already (both matchings in one launch, below)
        m, st = pre
        if st is not None:
          statuses.append(st)
      else:
        m, st = self._segm_match(iou.detach(), s_gt, 'lone')
        statuses.append(st)  # checked by the caller once the step has run (no host sync in here)
      cnt = torch.clamp(m.sum(dim=(1, 2)), min=1.0)
      if fixed:  # f_iou(pairwise=False) summed over ALL T, unmasked, over the identity match's count (full_model.py:922-945,985-1007)
        return (torch.diagonal(iou, dim1=1, dim2=2).sum(dim=1) / cnt).sum() / B, m
      return ((iou * m).sum(dim=(1, 2)) / cnt).sum() / B, m

    statuses = []
    # with the knob and use_iou_box the boxes are matched and scored on the stacked per-timestep corner IoUs
    iou_box_rows = iou_rows_stacked if batched else (torch.cat(iou_box_steps, dim=1).contiguous() if len(iou_box_steps) == T else None)
    # the two matchings are independent and each is one wave per image for milliseconds (dense soft-IoU
    # matrices early in training): the box matching runs on a side stream under the mask matching
    blf0, slf0 = opt.get('box_loss_fn', 'iou'), opt.get('segm_loss_fn', 'iou')
    if self.match_merged and self.fused_loss_head and not fixed and iou_box_rows is None and blf0 == 'iou' and slf0 == 'iou' and \
        B <= 256 and T <= 32:
      # the whole scalar head in one launch each way (LossHead): the IoU matrices, ONE launch for both matchings, the head
      want = ('iou_soft', 'inter', 'sum_a', 'sum_b')
      y_gt_c, box_gt_c = y_gt.contiguous(), box_gt.contiguous()
      st_s = ops.pair_stats(y_out.detach(), y_gt_c, want=want, a_tmajor=tm)
      st_b = ops.pair_stats(attn_box.detach(), box_gt_c, want=want, a_tmajor=tm)
      m2, st2 = self._segm_match(torch.cat([st_s['iou_soft'], st_b['iou_soft']], dim=0), torch.cat([s_gt, s_gt], dim=0), 'head')
      m, m_box = m2[:B], m2[B:]
      loss, pv = LossHead.apply(y_out, attn_box, s_out, y_gt_c, box_gt_c, st_s, st_b, m, m_box, tm, float(opt.get('loss_mix_ratio', 1.0)))
      pieces = {'loss': loss, 'box_loss': pv[0], 'segm_loss': pv[1], 'conf_loss': pv[2], 'iou_soft': pv[3], 'iou_soft_box': pv[4],
                'match': m, 'match_box': m_box, 'y_out': y_out.transpose(0, 1) if tm else y_out, 's_out': s_out, '_match_status': [st2]}
      return loss, pieces, stats
    if self.match_merged and not fixed and iou_box_rows is None:
      # The two Hungarian matchings are one wave per problem for milliseconds on the dense soft-IoU matrices of early
      # training: ONE launch over the 2 B problems (mask and box matrices side by side) instead of one matching under the
      # other on a side stream — no fork in the captured graph (31.5 -> 30.4 ms per step; starting the matchings early
      # from the sequential phase's masks on a side stream, under the stacked forward, loses to the fork it needs: 31.1,
      # and a second fork makes the graph replay 1.5x slower: 47.9 ms)
      i_soft, i_box = PairIoU.apply(y_out, y_gt, tm), PairIoU.apply(attn_box, box_gt, tm)
      m2, st2 = self._segm_match(torch.cat([i_soft.detach(), i_box.detach()], dim=0), torch.cat([s_gt, s_gt], dim=0), 'merged')
      iou_box, m_box = matched_iou(attn_box, box_gt, i_box, (m2[B:], None))
      iou_soft, m = matched_iou(y_out, y_gt, i_soft, (m2[:B], st2))
    elif self.match_side_stream:
      cur = torch.cuda.current_stream()
      side = _side_stream(dev)
      side.wait_stream(cur)
      with torch.cuda.stream(side):
        iou_box, m_box = matched_iou(attn_box, box_gt, iou_box_rows)
      iou_soft, m = matched_iou(y_out, y_gt)
      cur.wait_stream(side)
    else:
      iou_box, m_box = matched_iou(attn_box, box_gt, iou_box_rows)
      iou_soft, m = matched_iou(y_out, y_gt)
    box_loss, segm_loss = -iou_box, -iou_soft
    blf = opt.get('box_loss_fn', 'iou')
    if blf in ('mse', 'huber'):  # matched regression of (centre, log size) (full_model.py:891-892,952-964)
      import modellib
      gp, _ = ops.gt_box(y_gt, float(opt['attn_box_padding_ratio']), float(opt['padding']) + 4.0, want_box=False)
      ctr_gt, size_gt = (gp[:, :, 0:2] + gp[:, :, 2:4]) / 2.0, gp[:, :, 2:4] - gp[:, :, 0:2]
      params_gt = torch.cat([ctr_gt / (dims_hw / 2.0) - 1.0, torch.log(size_gt / dims_hw)], dim=2)
      params = torch.cat([cn_bt, ls_bt], dim=2) if batched else torch.cat([torch.stack(cn_list, dim=1), torch.stack(ls_list, dim=1)], dim=2)
      box_loss = modellib.f_match_loss(params, params_gt, m_box, T, modellib.f_squared_err if blf == 'mse' else modellib.f_huber)
    if opt.get('segm_loss_fn', 'iou') == 'wt_cov':  # modellib.f_weighted_coverage (modellib.py:292-302)
      iou_p = PairIoU.apply(y_out, y_gt, tm)
      sg = ops.pair_stats(y_gt, y_gt, want=('sum_b',))['sum_b']
      wts = sg / (sg.sum(dim=1, keepdim=True) + (sg == 0).to(sg.dtype))
      segm_loss = -(iou_p.max(dim=1)[0] * wts).sum() / B
    s_min = torch.cummin(s_out, dim=1)[0]
    s_max = torch.flip(torch.cummax(torch.flip(s_out, [1]), dim=1)[0], [1])
    ms = m.sum(dim=2)
    conf = (-ms * torch.log(s_min + 1e-5) - (1 - ms) * torch.log(1 - s_max + 1e-5)).sum() / B / T
    loss = box_loss + segm_loss + float(opt.get('loss_mix_ratio', 1.0)) * conf
    pieces = {'loss': loss, 'box_loss': box_loss, 'segm_loss': segm_loss, 'conf_loss': conf, 'iou_soft': iou_soft,
              'iou_soft_box': iou_box, 'match': m, 'match_box': m_box, 'y_out': y_out.transpose(0, 1) if tm else y_out, 's_out': s_out,
              '_match_status': statuses}
    return loss, pieces, stats

  # ------------------------------------------------------------------ one optimisation step
  use_graph = True  # capture forward + backward + EMA of a repeated step shape in one HIP graph
  prepack = os.environ.get('RA_PREPACK', '1') != '0'    # every filter packed / bias padded by ONE gather per step (_PackCache); False: one launch per layer
  own_gemm = os.environ.get('RA_OWN_GEMM', '1') != '0'   # the controller's parameter-gradient products on ra_gemm_tn_acc_f32; False: torch.mm / addmm / addmv

  def _grads_and_stats(self, x, y_gt, s_gt, knobs, generator, extra):
    """zero_grad + forward + backward + BN EMA update: everything of a step that is pure device work."""
    self.bucket.zero_grad()
    loss, pieces, stats = self.forward_loss(x, y_gt, s_gt, knobs=knobs, generator=generator, **extra)
    loss.backward()
    with torch.no_grad():  # shadow = 0.9 shadow + 0.1 batch statistic (nnlib.py:103-110)
      if set(stats.keys()) == set(self._stat_views.keys()):  # every BN copy ran: one update of the flat buffers
        self.ema.mul_(EMA_DECAY).add_(self.stat, alpha=1 - EMA_DECAY)
      else:
        for key, (mean, var) in stats.items():
          self.model[key + '_ema_mean'].mul_(EMA_DECAY).add_((1 - EMA_DECAY) * mean)
          self.model[key + '_ema_var'].mul_(EMA_DECAY).add_((1 - EMA_DECAY) * var)
    return {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in pieces.items()}

  # f_segm_match's Hungarian problems on host cores, side by side, as a host node of the captured step (round 6): the device
  # solver's launch lasts as long as its slowest problem — 1.7 ms of a cfg4 step, with every gradient waiting for it — where a
  # host core needs ~0.5 ms per problem.  RA_HUNG_HOST=0: the device solver (ra_segm_match_f32)
  host_match = os.environ.get('RA_HUNG_HOST', '1') != '0'

  def _segm_match(self, iou, s_gt, site):
    if not self.host_match:
      return ops.segm_match(iou, s_gt)
    blocks = self.__dict__.setdefault('_match_blocks', {})
    # one block per (call site, shape, position in the step): the box matching may run on a side stream under the mask matching
    # (match_side_stream) — two calls in flight at once must not share the staging block and its control words.  The position
    # counts the step's matching calls (forward_loss resets it), so the eager first step and the captured one find the same
    # blocks (a block made DURING capture would pin memory inside it and invalidate the capture)
    seq = self.__dict__.get('_match_seq', 0)
    self._match_seq = seq + 1
    key = (site, tuple(iou.shape), seq)
    threads = getattr(self, '_match_threads', None)
    if threads is None:
      n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
      threads = self._match_threads = max(1, min(32, n))
    m, st, blocks[key] = ops.segm_match_host(iou, s_gt, blocks.get(key), threads)  # the block lives as long as the trainer (and its graphs)
    return m, st

  def _check_status(self, rec):
    """The status words of one finished step, on the host.  The guarded optimizer kernel has already refused that step's
    update if any of them was non-zero, so nothing here repairs weights.  The record is
      [n_match matching codes | n_ctrl controller words | n_forced "skipped on purpose" | world > 1: 2 all-reduced flags]
    and the decision is COLLECTIVE: the two trailing flags — "a matching failed on some rank", "a controller
    timed out (or the step was skipped on purpose) on some rank" — rode on the gradient bucket's all-reduce, so every rank
    reads the same two words and takes the same branch:
      * a failed MATCHING anywhere raises on every rank (the reference aborts there, hungarian.cc LOG(FATAL));
      * a controller TIME-OUT anywhere (the 16-workgroup form found a peer not resident: something else on the GPU) makes
        every rank switch its sequential phase to the one-workgroup controller for good, give the skipped step's number back
        and warn — no rank walks into the next collective alone, global_step and the learn-rate schedule stay equal;
      * a step that run() skipped on purpose (the one already queued behind a timed-out step: it ran on the exchange
        workspace that step left behind) only gives its number back.
    Returns True when a recovery ran (run() then forces the skip of the step it has just launched)."""
    ev, host, n_match, n_ctrl = rec[:4]
    n_forced = rec[4] if len(rec) > 4 else 0
    n_world = rec[5] if len(rec) > 5 else max(0, int(host.numel()) - n_match - n_ctrl - n_forced)
    ev.synchronize()
    h = host.clone()
    o = n_match + n_ctrl
    if n_forced and int(h[o:o + n_forced].abs().max()) != 0:  # set together on every rank (the decision that set it was collective)
      self.bucket.global_step = max(0, self.bucket.global_step - 1)
      self.model['global_step'] = float(self.bucket.global_step)
      self.skipped_steps = getattr(self, 'skipped_steps', 0) + 1
      return False
    w = h[o + n_forced:o + n_forced + n_world]
    own_ctrl = bool(n_ctrl and int(h[n_match:o].abs().max()) != 0)
    if n_match:
      ops.check_match_status(h[:n_match], 'f_segm_match')  # raises with the solver's own code
    if n_world >= 2:
      any_match, any_ctrl = int(w[0]) != 0, int(w[1]) != 0
    else:  # one process — or a record from before the flags were split: any foreign flag is a failure we cannot name
      any_match, any_ctrl = bool(n_world and int(w.abs().max()) != 0), False
    if any_match:
      raise rn.RecAttendError('training step: another rank reported a failed matching for this step; the update was skipped '
                              'on every rank')
    if own_ctrl or any_ctrl:  # a controller workgroup timed out on its peers, here or on another rank
      sc = getattr(self, '_seqc', None)
      if sc is not None and sc.get('status') is not None:
        sc['status'].zero_()
      import warnings
      warnings.warn('controller_split (sequential phase of the training step): a workgroup waited for a peer that never became '
                    'resident%s — is another process using this GPU?  That step\'s update was not applied on any rank; the '
                    'trainer runs the one-workgroup controller from now on (RA_TRAIN_CTRL_SPLIT=0 selects it from the start)'
                    % ('' if own_ctrl else ' on another rank'))
      self.seq_ctrl_split = False
      self._seqc = None
      self._drop_captured_steps()
      self.bucket.global_step = max(0, self.bucket.global_step - 1)  # the skipped step is taken again
      self.model['global_step'] = float(self.bucket.global_step)
      self.skipped_steps = getattr(self, 'skipped_steps', 0) + 1
      return True
    return False

  def flush_status(self):
    """Check the solver / controller statuses of the LAST step now (run() checks each step's record one step late)."""
    rec = getattr(self, '_status_pending', None)
    self._status_pending = None
    if rec is not None:
      self._check_status(rec)

  def _static_inputs(self, x, y_gt, s_gt, knobs, extra):
    """The captured step's static input buffers for these shapes ({x, y, d, c} as image_ops names them), or None: the
    augmentation then writes into them and _graphed finds nothing to copy."""
    use_knob = bool(self.opt.get('use_knob', False)) or isinstance(self, BoxTrainStep)
    shp = {'x': x, 'y_gt': y_gt, 's_gt': s_gt}
    shp.update({k: v for k, v in extra.items() if v is not None})
    try:
      ins = tuple((k, tuple(np.shape(v) if not isinstance(v, torch.Tensor) else v.shape)) for k, v in sorted(shp.items()))
      ks = {k: tuple(v.shape) for k, v in knobs.items()} if knobs is not None else (self.knob_shapes(np.shape(x)[0]) if use_knob else {})
    except Exception:
      return None
    st = self._graphs.get(ins + tuple(sorted(ks.items())))
    if st is None or 'graph' not in st:
      return None
    return {'x': st['ins']['x'], 'y': st['ins']['y_gt'], 'd': st['ins'].get('d_in'), 'c': st['ins'].get('y_in')}

  def _graphed(self, x, y_gt, s_gt, knobs, generator, extra):
    """The same work as _grads_and_stats, replayed from a HIP graph.  A training step issues ~24 000
    kernels (16 timesteps x 40 layers x forward / backward pieces plus the dense glue under autograd);
    eagerly the host needs ~10 us for each, more than most of them run.  Inputs, the step's random
    draws and the two knob probabilities live in static device buffers that are refreshed before
    every replay; the first step of a shape runs eagerly (it also warms every cache), the second one
    is captured."""
    dev = self.bucket.param.device
    as_t = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float32))).to(
        device=dev, dtype=torch.float32)
    ins = {'x': as_t(x), 'y_gt': as_t(y_gt), 's_gt': as_t(s_gt)}
    ins.update({k: as_t(v) for k, v in extra.items() if v is not None})
    use_knob = bool(self.opt.get('use_knob', False)) or isinstance(self, BoxTrainStep)
    in_place = False
    if knobs is None and use_knob:
      # a captured step of this shape exists: draw straight into its static buffers
      kkey = tuple((k, tuple(v.shape)) for k, v in sorted(ins.items())) + tuple(sorted(self.knob_shapes(ins['x'].shape[0]).items()))
      st0 = self._graphs.get(kkey)
      if st0 is not None and 'graph' in st0:
        knobs, in_place = self.draw_knobs(ins['x'].shape[0], generator, out=st0['knobs']), True
      else:
        knobs = self.draw_knobs(ins['x'].shape[0], generator)
    knobs = {k: as_t(v) for k, v in (knobs or {}).items()}
    key = tuple((k, tuple(v.shape)) for k, v in sorted(ins.items())) + tuple((k, tuple(v.shape)) for k, v in sorted(knobs.items()))
    st = self._graphs.get(key)
    if st is None:  # first step of this shape: eager
      self._graphs[key] = {'warm': True}
      self._sched = None
      return self._grads_and_stats(ins['x'], ins['y_gt'], ins['s_gt'], knobs or None, None,
                                   {k: ins[k] for k in ('d_in', 'y_in') if k in ins})
    step = self.bucket.global_step
    sched = [knob_prob(self.opt, step, self.opt['knob_box_offset']), knob_prob(self.opt, step, self.opt['knob_segm_offset'])] \
        if bool(self.opt.get('use_knob', False)) else [0.0, 0.0]
    if 'graph' not in st:
      st['ins'] = {k: v.clone() for k, v in ins.items()}
      st['knobs'] = {k: v.clone() for k, v in knobs.items()}
      st['sched'] = torch.tensor(sched, dtype=torch.float32, device=dev)
      self._sched = st['sched']
      self._pack.merge()  # the eager first step met every filter geometry: the captured step packs them with one gather
      torch.cuda.synchronize()
      g = torch.cuda.CUDAGraph()
      with rn.quiet_capture(), torch.cuda.graph(g):
        st['out'] = self._grads_and_stats(st['ins']['x'], st['ins']['y_gt'], st['ins']['s_gt'], st['knobs'] or None, None,
                                          {k: st['ins'][k] for k in ('d_in', 'y_in') if k in st['ins']})
      self._sched = None
      st['graph'] = g
    else:
      for k, v in ins.items():
        if v.data_ptr() != st['ins'][k].data_ptr():  # the augmentation wrote into the static buffer
          st['ins'][k].copy_(v)
      if not in_place:
        for k, v in knobs.items():
          st['knobs'][k].copy_(v)
      # pinned and asynchronous: a pageable source made this copy wait for the PREVIOUS step on the stream — a host sync
      # per step with the GPU idle behind it.  Four buffers in rotation, each with the event of the copy that last read it.
      ring = st.setdefault('sched_host', [torch.empty(2, dtype=torch.float32).pin_memory() for _ in range(4)])
      evs = st.setdefault('sched_ev', [None] * len(ring))
      st['sched_i'] = (st.get('sched_i', -1) + 1) % len(ring)
      hb = ring[st['sched_i']]
      if evs[st['sched_i']] is not None:  # the copy that last read this buffer has run (with no status words to check —
        evs[st['sched_i']].synchronize()  # fixed_order and the one-workgroup controller — nothing else holds the host back)
      hb[0], hb[1] = float(sched[0]), float(sched[1])
      st['sched'].copy_(hb, non_blocking=True)
      evs[st['sched_i']] = torch.cuda.Event()
      evs[st['sched_i']].record()
    st['graph'].replay()
    return dict(st['out'])

  def augment(self, x, y_gt, extra, aug=None, out=None):
    """The in-graph augmentation in front of the training graph (full_model.py:203-232, box_model.py: the same
    call): img.random_transformation on x, y_gt[, d_in, y_in] with phase_train true — ONE crop offset in
    [0, 2 * padding) per batch after zero padding, and the flip / transpose decisions model_opt switches on
    (rnd_hflip / rnd_vflip / rnd_transpose; the reference's trainers leave them off, full_model_train.py:653-656).
    aug: None = draw from this trainer's (rank-offset) generator; a dict = given draws; False = none (the identity
    crop, what evaluation does)."""
    if aug is False:
      return x, y_gt, extra
    import image_ops
    dev = self.bucket.param.device
    as_t = lambda a: None if a is None else (a if isinstance(a, torch.Tensor) else torch.as_tensor(
        np.asarray(a, dtype=np.float32))).to(device=dev, dtype=torch.float32).contiguous()
    opt = self.opt
    r = image_ops.random_transformation(
        as_t(x), int(opt.get('padding', 0)), True, rnd_vflip=bool(opt.get('rnd_vflip', False)),
        rnd_hflip=bool(opt.get('rnd_hflip', False)), rnd_transpose=bool(opt.get('rnd_transpose', False)),
        rnd_colour=bool(opt.get('rnd_colour', False)), y=as_t(y_gt),
        d=as_t(extra.get('d_in')), c=as_t(extra.get('y_in')), generator=self.aug_gen, draws=aug if isinstance(aug, dict) else None,
        out=out)
    extra = dict(extra)
    if 'd' in r:
      extra['d_in'] = r['d']
    if 'c' in r:
      extra['y_in'] = r['c']
    self.last_aug = r.get('_draws')
    return r['x'], r['y'], extra

  def run(self, x, y_gt, s_gt, knobs=None, generator=None, aug=None, **extra):
    """loss + train_step: augmentation, backward into the flat bucket, one all-reduce, clip + Adam, BN EMA update.
    The reported `loss` excludes nothing the reference includes except the weight-decay terms, which
    enter through their gradient (wd * w) inside the optimizer kernel."""
    if knobs is not None:
      dev = self.bucket.param.device
      knobs = {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, dtype=np.float32))).to(
          device=dev, dtype=torch.float32) for k, v in knobs.items()}
    x, y_gt, extra = self.augment(x, y_gt, extra, aug, self._static_inputs(x, y_gt, s_gt, knobs, extra) if self.use_graph else None)
    if self.use_graph:
      out = self._graphed(x, y_gt, s_gt, knobs, generator, extra)
    else:
      self._sched = None
      out = self._grads_and_stats(x, y_gt, s_gt, knobs, generator, extra)
    # The solver / controller statuses of this step gate its update ON THE DEVICE (GradBucket.step(status=...): the optimizer
    # kernel leaves parameters and moments alone if any word is non-zero) and go to pinned host memory behind the step on the
    # stream; what the host CHECKS here is the previous step's record — it waits for that step to finish while this step's
    # graph is already queued behind it, so the GPU does not idle — and it does so BEFORE this step's all-reduce and
    # optimizer launch are issued: a failed step k raises (or recovers) with the weights exactly as step k found them.
    # flush_status() checks the record of the last step (full_model.run with numpy outputs, checkpointing, the bench).
    sts = [st.reshape(-1).to(torch.int32) for st in out.pop('_match_status', [])]
    sc = getattr(self, '_seqc', None)
    n_match = sum(int(t.numel()) for t in sts)
    n_ctrl = 0
    if sc is not None and sc.get('ok'):
      # a SNAPSHOT, taken on the stream behind this step's launches: .reshape / .to(int32) of an int32 tensor alias the
      # controller's sticky status word itself, which the check of the PREVIOUS step below may zero
      sts.append(sc['status'].reshape(-1).to(torch.int32).clone())
      n_ctrl = int(sts[-1].numel())
    inject = getattr(self, '_inject_status', None)  # test hook: a status word forced for ONE step
    if inject is not None and sts:
      self._inject_status = None
      sts[0] = sts[0] + int(inject)
    inject_c = getattr(self, '_inject_ctrl_status', None)  # test hook: "this step's controller timed out"
    if inject_c is not None and n_ctrl:
      self._inject_ctrl_status = None
      sts[-1] = sts[-1] + int(inject_c)
    prev = getattr(self, '_status_pending', None)
    self._status_pending = None
    recovered = False
    if prev is not None:
      recovered = bool(self._check_status(prev))  # also before its pinned buffer is overwritten by this step's copy
    n_forced = 0
    if recovered and sts:
      # the step just launched was queued behind the timed-out one: it ran the 16-workgroup controller on the exchange
      # workspace that step left behind (generation tags half advanced), so its gradients are not to be trusted either —
      # its update is refused on the device like its predecessor's and its step number is given back at its own check
      sts.append(torch.ones(1, dtype=torch.int32, device=sts[0].device))
      n_forced = 1
    dev_st = None
    if sts:
      dev_st = (torch.cat(sts) if len(sts) > 1 else sts[0]).contiguous()
      if self.world > 1:  # every rank must take the same decision: two flags ride on the bucket's all-reduce (sums over ranks)
        self.bucket.grad_full[self.bucket.n] = (dev_st[:n_match] < 0).any().to(torch.float32) if n_match else 0.0
        self.bucket.grad_full[self.bucket.n + 1] = (dev_st[n_match:] != 0).any().to(torch.float32) if dev_st.numel() > n_match else 0.0
    world = self.bucket.allreduce()
    n_world = 0
    if dev_st is not None and world > 1:
      dev_st = torch.cat([dev_st, self.bucket.grad_full[self.bucket.n:self.bucket.n + 2].view(torch.int32)])
      n_world = 2
    lr = self.bucket.step(world=world, status=dev_st, n_solver=n_match)
    if dev_st is not None:
      host = getattr(self, '_status_host', None)
      if host is None or host.numel() != dev_st.numel():
        host = self._status_host = torch.empty(dev_st.numel(), dtype=torch.int32).pin_memory()
      ev = torch.cuda.Event()
      host.copy_(dev_st, non_blocking=True)
      ev.record()
      self._status_pending = (ev, host, n_match, n_ctrl, n_forced, n_world)
    wd = float(self.opt.get('weight_decay', 0.0) or 0.0)
    if self.use_graph:  # the graph's output tensors are rewritten by the next replay: hand out copies of the small ones
      out = {k: (v.clone() if isinstance(v, torch.Tensor) and v.numel() <= 4096 else v) for k, v in out.items()}
    out['learn_rate'] = lr
    out['weight_decay_loss'] = 0.5 * (self.bucket.wd * self.bucket.param * self.bucket.param).sum().detach() if wd else 0.0
    return out


class BoxTrainStep(TrainStep):
  """box_model's training graph (box_model.py:403-652): the controller-only model whose canvas is
  always teacher-forced from the greedily matched ground truth times (1 - U[0, 0.3)) noise; loss =
  matched box loss ('iou' | 'mse' | 'huber') + conf loss (sigmoid score, or 1 - softmax[:, :, 0] with
  several semantic classes).  Pre-trains the controller weights full_model picks up
  (run_cvppp.sh:16-28).  Same kernels and bucket as TrainStep."""

  def __init__(self, model, world=1):
    self.model, self.opt, self.d = model, model.opt, model.dims
    if not torch.cuda.is_available():
      raise rn.RecAttendError('the training step needs an MI355X (HIP device); there is no CPU fallback')
    if self.opt.get('box_loss_fn', 'iou') not in ('iou', 'mse', 'huber'):
      raise NotImplementedError("box_loss_fn in ('iou', 'mse', 'huber')")
    self._setup(model, world)
    # the controller CNN reads concat(x, canvas[, d_in, y_in]) (box_model.py:404-410): packed to C0p channels, the
    # first filter's rows found through the channel map (stage 1 of run_kitti.sh:45-59 trains with --add_d_out --add_y_out)
    cmap_c, _ = model.engine._chan_map(self.d['ctrl_in'])
    self.cmap_c = None if cmap_c == list(range(len(cmap_c))) else cmap_c
    self.cmap_a = None

  def draw_knobs(self, B, generator=None, out=None):
    """The step's one random draw: the canvas noise U[0, 0.3) (box_model.py:500-502); out: drawn into the captured
    step's static buffer."""
    d = self.d
    if out is not None:
      out['noise'].uniform_(0.0, 0.3, generator=generator)
      return out
    return {'noise': 0.3 * torch.rand((d['T'], B, d['H'], d['W']), generator=generator, device=self.bucket.param.device)}

  def knob_shapes(self, B):
    d = self.d
    return {'noise': (d['T'], B, d['H'], d['W'])}

  def forward_loss(self, x, y_gt, s_gt, knobs=None, generator=None, d_in=None, y_in=None):
    P, d, opt = self.leaves, self.d, self.opt
    self._pack.clear()  # the optimizer wrote new weights since the last step
    self._match_seq = 0  # (_segm_match's staging blocks are keyed by the call's position in the step)
    self._pack_epoch = getattr(self, '_pack_epoch', 0) + 1
    self._pack.refresh()
    self._wgrad_parts.reset()
    if getattr(self, '_ctl', None) is not None:
      self._ctl.begin_step()
    dev = self.bucket.param.device
    as_t = lambda a: (a if isinstance(a, torch.Tensor) else torch.as_tensor(np.asarray(a, dtype=np.float32))).to(
        device=dev, dtype=torch.float32).contiguous()
    x, y_gt, s_gt = as_t(x), as_t(y_gt), as_t(s_gt)
    extra = []
    if d['add_d_out']:  # box_model.py:78-131: either both or neither
      if d_in is None or y_in is None:
        raise rn.RecAttendError('this architecture feeds d_in and y_in (box_model.py:78-131)')
      extra = [as_t(d_in), as_t(y_in)]
    B, T, H, W, Fh, Fw = x.shape[0], d['T'], d['H'], d['W'], d['Fh'], d['Fw']
    noise = as_t(knobs['noise']) if knobs is not None and 'noise' in knobs else self.draw_knobs(B, generator)['noise']
    fixed = bool(opt.get('fixed_order', False))
    gp, box_gt = ops.gt_box(y_gt, float(opt['attn_box_padding_ratio']), 10.0)   # get_gt_attn's default min_padding
    canvas = torch.zeros((B, H, W, 1), device=dev)
    ysel = torch.empty((B, H, W), device=dev)
    stats, box_list, s_list, cn_list, ls_list = {}, [], [], [], []
    dims_hw = _const('dims', (H, W), dev, lambda: torch.tensor([H, W], dtype=torch.float32, device=dev))
    head_flags = (1 if d['squash'] else 0) | (2 if d['fixed_var'] else 0) | (4 if d['dynamic_var'] else 0) | 8
    for tt in range(T):
      inp = torch.cat([x, canvas] + extra, dim=3)   # packed [x | canvas | d_in | y_in], zero-padded to C0p
      if inp.shape[3] != d['C0p']:
        inp = _pad_channels(inp)
      feat = self._cnn(inp, 'ctrl_cnn', d['ccnn_nlayers'], d['ccnn_pool'], tt, self.cmap_c, stats)[-1]
      h, co = self._controller(feat.reshape(B, d['G'], -1), tt)
      cn, ls, ctr, size, lg_var, _, bgm, _, _ = AttnHead.apply(co, H, W, Fh, Fw, head_flags)
      box = AttnPaste.apply(None, ctr, size, lg_var, bgm, H, W, Fh, Fw)
      if fixed:
        gsel = y_gt[:, tt]
      else:
        iou_t = ops.pair_stats(box.detach()[:, None].contiguous(), box_gt, want=('iou_soft',))['iou_soft']
        ops.weighted_sum(ops.greedy_match(iou_t.view(B, T)), y_gt, ysel)
        gsel = ysel
      gsel = gsel - gsel * noise[tt]
      canvas = torch.maximum(gsel[..., None], canvas)                            # stop_gradient (:504)
      s = self._linear(h, 'score_mlp_w_0', 'score_mlp_b_0')
      s_list.append((torch.sigmoid(s) if d['nsc'] == 1 else torch.softmax(s, dim=1))[:, None])
      box_list.append(box)
      cn_list.append(cn)
      ls_list.append(ls)
    attn_box, s_out = torch.stack(box_list, dim=1), torch.cat(s_list, dim=1)
    iou = PairIoU.apply(attn_box, box_gt)
    statuses = []
    if fixed:
      m = torch.eye(T, device=dev)[None] * s_gt[:, None, :] * s_gt[:, :, None]
    else:
      m, st = self._segm_match(iou.detach(), s_gt, 'box')
      statuses.append(st)
    cnt = torch.clamp(m.sum(dim=(1, 2)), min=1.0)
    iou_box = ((iou * m).sum(dim=(1, 2)) / cnt).sum() / B
    box_loss = -iou_box
    blf = opt.get('box_loss_fn', 'iou')
    if blf in ('mse', 'huber'):
      import modellib
      ctr_gt, size_gt = (gp[:, :, 0:2] + gp[:, :, 2:4]) / 2.0, gp[:, :, 2:4] - gp[:, :, 0:2]
      params_gt = torch.cat([ctr_gt / (dims_hw / 2.0) - 1.0, torch.log(size_gt / dims_hw)], dim=2)
      params = torch.cat([torch.stack(cn_list, dim=1), torch.stack(ls_list, dim=1)], dim=2)
      box_loss = modellib.f_match_loss(params, params_gt, m, T, modellib.f_squared_err if blf == 'mse' else modellib.f_huber)
    sc = s_out[:, :, 0] if d['nsc'] == 1 else 1.0 - s_out[:, :, 0]               # box_model.py:620-625
    s_min = torch.cummin(sc, dim=1)[0]
    s_max = torch.flip(torch.cummax(torch.flip(sc, [1]), dim=1)[0], [1])
    ms = m.sum(dim=2)
    conf = (-ms * torch.log(s_min + 1e-5) - (1 - ms) * torch.log(1 - s_max + 1e-5)).sum() / B / T
    loss = box_loss + conf
    pieces = {'loss': loss, 'box_loss': box_loss, 'conf_loss': conf, 'iou_soft_box': iou_box, 'match_box': m,
              's_out': s_out[:, :, 0] if d['nsc'] == 1 else s_out, 'attn_box': attn_box, '_match_status': statuses}
    return loss, pieces, stats
