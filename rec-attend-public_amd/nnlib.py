"""Operator library with the reference's `nnlib` surface, on hand-written gfx950 kernels.

Same constructor names, argument lists, return structure and `model[...]` key names as the
reference's nnlib.py (cited per function as nnlib.py:line), but eager: tensors are float32
torch CUDA tensors in NHWC, and each closure launches HIP kernels from librecattend.so
through the C ABI (ra_ops).  Constructors register their weights into `model` exactly like
the reference (`'{scope}_w_{i}'`, `'{scope}_{i}_{copy}_{beta,gamma,ema_mean,ema_var}'`, ...).

`phase_train=False`: the decode loop's fused eval kernels (EMA statistics folded into the conv
epilogue).  `phase_train=True` (round 5): the training step's kernels behind the same closures —
each cnn / dcnn layer is one `ra_train.ConvBNActPool` node (conv with the batch moments in its
epilogue, BN + ReLU + pool on the batch statistics, MFMA backward-data / backward-weight, the
statistics differentiated through), `batch_norm` is `ra_train.BatchNormTrain`, and every call
moves the copy's EMA shadows by `shadow = decay shadow + (1 - decay) value`, `decay = 1 - 0.1
phase_train` (nnlib.py:103-110).  Gradients are torch autograd's: mark the registered tensors
(`model[...]`) with `requires_grad_()` and differentiate the closures' outputs — the counterpart
of `tf.gradients` over the reference graph.  The optimisation step of full_model (`ra_train.TrainStep`)
runs the same kernels without going through these closures.
"""
import numpy as np
import torch

import ra_ops as ops
from ra_native import RecAttendError

_collections = {'losses': []}  # (weight tensor, wd) pairs; nnlib.py:59-61


def device():
  return torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')


def get_collection(name):
  return _collections.setdefault(name, [])


def _is_train(phase_train):
  if phase_train is None:
    return False
  if isinstance(phase_train, torch.Tensor):
    return bool(phase_train.item())
  if isinstance(phase_train, (list, dict)):  # a mutable holder {'value': bool}
    return bool(phase_train['value'] if isinstance(phase_train, dict) else phase_train[0])
  return bool(phase_train)


EMA_DECAY_TRAIN = 0.9  # decay = 1 - 0.1 * phase_train (nnlib.py:103-104)


def _ema_update(shadow_mean, shadow_var, mean, var):
  """ema.apply([batch_mean, batch_var]) of a training call (nnlib.py:105-110): the copy's shadows move towards the
  batch statistics; not differentiated (the shadows are assigned, nnlib.py:121-127)."""
  with torch.no_grad():
    shadow_mean.mul_(EMA_DECAY_TRAIN).add_(mean, alpha=1.0 - EMA_DECAY_TRAIN)
    shadow_var.mul_(EMA_DECAY_TRAIN).add_(var, alpha=1.0 - EMA_DECAY_TRAIN)


def truncated_normal_initializer(stddev=0.01, seed=None):
  """tf.truncated_normal_initializer: N(0, stddev) re-drawn outside 2 stddev (nnlib.py:53-54)."""
  gen = None
  if seed is not None:
    gen = torch.Generator().manual_seed(seed)

  def init(shape):
    t = torch.empty(tuple(shape), dtype=torch.float32)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=stddev, a=-2 * stddev, b=2 * stddev,
                                generator=gen)
    return t

  return init


def constant_initializer(value):
  return lambda shape: torch.full(tuple(shape), float(value), dtype=torch.float32)


def weight_variable(shape, initializer=None, init_val=None, wd=None, name=None, trainable=True):
  """nnlib.py:41-62.  Returns a float32 tensor on the compute device."""
  if initializer is None:
    initializer = truncated_normal_initializer(stddev=0.01)
  if init_val is None:
    var = initializer(shape)
  else:
    var = torch.as_tensor(np.asarray(init_val, dtype=np.float32)).reshape(tuple(shape)).clone()
  var = var.to(device()).contiguous()
  var.ra_name = name
  var.ra_trainable = trainable
  if wd:
    get_collection('losses').append((var, wd))
  return var


def conv2d(x, w, stride=1):
  """nnlib.py:6-12 — 3x3, stride-1 'SAME' cross-correlation; x [B,H,W,D], w [F,F,In,Out]."""
  if stride != 1 or tuple(w.shape[:2]) != (3, 3):
    raise RecAttendError('conv2d: only 3x3 stride-1 is built (the only case on the path)')
  cin = x.shape[3]
  cin_k = -(-cin // 4) * 4
  if cin_k != cin:
    x = torch.nn.functional.pad(x, (0, cin_k - cin)).contiguous()
  wp = torch.from_numpy(ops.pack_conv_weights(w, cin_kernel=cin_k)).to(x.device)
  sc, sh = ops.fold_bn(None, w.shape[3])
  return ops.conv3x3(x.contiguous(), wp, torch.from_numpy(sc).to(x.device),
                     torch.from_numpy(sh).to(x.device), w.shape[3], relu=False, pool=1)


def max_pool(x, ratio):
  """nnlib.py:15-25."""
  return ops.max_pool(x.contiguous(), ratio)


def relu(x):
  C = x.shape[-1]
  one = torch.ones(C, dtype=torch.float32, device=x.device)
  return ops.affine_act(x.contiguous(), one, torch.zeros_like(one), relu=True)


def _bn_register(model, scope2, n_out, init_beta=None, init_gamma=None):
  """Creates (or finds) the four BN tensors of one copy.  EMA shadows start at 0
  (tf.train.ExponentialMovingAverage over tensors; nnlib.py:104-116)."""
  out = []
  for name, init in (('beta', init_beta), ('gamma', init_gamma), ('ema_mean', None),
                     ('ema_var', None)):
    key = '{}_{}'.format(scope2, name)
    if model is not None and key in model:
      out.append(model[key])
      continue
    if init is None:
      init = np.full([n_out], 1.0 if name == 'gamma' else 0.0, dtype=np.float32)
    t = weight_variable([n_out], init_val=init, name=name)
    if model is not None:
      model[key] = t
    out.append(t)
  return out


def batch_norm(x, n_out, phase_train, scope='bn', scope2='bn', affine=True, init_beta=None,
               init_gamma=None, frozen=False, model=None):
  """nnlib.py:65-128.  Eval: gamma*(x-ema_mean)/sqrt(ema_var+1e-3)+beta.  Training: the same with the batch moments
  over (B,H,W) (biased variance, differentiated through), and the EMA shadows updated (nnlib.py:98-112)."""
  beta, gamma, mean, var = _bn_register(model, scope2, n_out, init_beta, init_gamma)
  if _is_train(phase_train):
    import ra_train as rt
    y, bmean, bvar = rt.BatchNormTrain.apply(x, gamma, beta)
    _ema_update(mean, var, bmean, bvar)
    return y
  sc, sh = ops.fold_bn(None, n_out, (beta, gamma, mean, var))
  return ops.affine_act(x.contiguous(), torch.from_numpy(sc[:n_out].copy()).to(x.device),
                        torch.from_numpy(sh[:n_out].copy()).to(x.device))


class _LayerCache(object):
  """Packed weights / folded BN per (layer, copy), rebuilt when a source tensor changes."""

  def __init__(self):
    self.store = {}

  def get(self, key, tensors, build):
    stamp = tuple((t.data_ptr(), t._version) for t in tensors if t is not None)
    hit = self.store.get(key)
    if hit is None or hit[0] != stamp:
      hit = (stamp, build())
      self.store[key] = hit
    return hit[1]


def _dev(a, like):
  return torch.from_numpy(np.ascontiguousarray(a)).to(like.device)


def cnn(f, ch, pool, act, use_bn, phase_train=None, wd=None, scope='cnn', model=None,
        init_weights=None, frozen=None, shared_weights=None):
  """nnlib.py:131-257.  Returns run_cnn(x) -> list of the N layer outputs.

  Each layer is ONE fused kernel: conv3x3 + bias + BN(copy) + ReLU + max-pool.  `act[i]`
  must be relu (or None); `f[i]` must be 3.  BN parameters are separate per call ("copy"),
  like the reference's copy counter (nnlib.py:212,254); run_cnn.reset_copy() rewinds it for
  the next forward pass, run_cnn(x, copy=t) addresses a copy explicitly."""
  nlayers = len(f)
  w = [None] * nlayers
  b = [None] * nlayers
  for ii in range(nlayers):
    if f[ii] != 3:
      raise RecAttendError('cnn: filter size %d not built (path uses 3x3 only)' % f[ii])
    iw = init_weights[ii] if init_weights is not None and init_weights[ii] is not None else None
    trainable = not (frozen is not None and frozen[ii])
    if shared_weights:
      w[ii], b[ii] = shared_weights[ii]['w'], shared_weights[ii]['b']
    else:
      w[ii] = weight_variable([f[ii], f[ii], ch[ii], ch[ii + 1]], name='w',
                              init_val=None if iw is None else iw['w'], wd=wd,
                              trainable=trainable)
      b[ii] = weight_variable([ch[ii + 1]], init_val=None if iw is None else iw['b'],
                              initializer=None, name='b', trainable=trainable)
    if model is not None:
      for name, param in zip(['w', 'b'], [w[ii], b[ii]]):
        key = '{}_{}_{}'.format(scope, name, ii)
        if key in model:
          raise Exception('Key exists: {}'.format(key))
        model[key] = param
  copy = [0]
  cache = _LayerCache()
  relu_fn = relu

  def _bn(ii, cp):
    iw = init_weights[ii] if init_weights is not None and init_weights[ii] is not None else None
    ib = None if iw is None else iw.get('beta_{}'.format(cp))
    ig = None if iw is None else iw.get('gamma_{}'.format(cp))
    return _bn_register(model if model is not None else run_cnn.local, '{}_{}_{}'.format(
        scope, ii, cp), ch[ii + 1], ib, ig)

  def declare_copies(n):
    """Create the BN tensors of copies 0..n-1 up front (the reference has them after graph
    construction: one copy per timestep)."""
    for ii in range(nlayers):
      if use_bn[ii]:
        for cp in range(n):
          _bn(ii, cp)

  def layer_params(ii, cp, x_like, cin_kernel=None, chan_map=None):
    """(packed weights, scale, shift) device tensors of layer ii / BN copy cp."""
    cin_k = ch[ii] if cin_kernel is None else cin_kernel
    cin_k = -(-cin_k // 4) * 4
    wp = cache.get(('w', ii, cin_k, None if chan_map is None else tuple(chan_map)), [w[ii]],
                   lambda: _dev(ops.pack_conv_weights(w[ii], cin_kernel=cin_k,
                                                      chan_map=chan_map), x_like))
    bn = _bn(ii, cp) if use_bn[ii] else None
    srcs = [b[ii]] + (list(bn) if bn else [])
    sc, sh = cache.get(('bn', ii, cp), srcs, lambda: tuple(
        _dev(a, x_like) for a in ops.fold_bn(b[ii], ch[ii + 1], bn)))
    return wp, sc, sh

  def run_cnn_train(x, cp):
    """phase_train = True: per layer conv + b -> BN on the batch moments -> ReLU -> max-pool as one autograd node on the
    training step's kernels; the copy's EMA shadows move (nnlib.py:229-253 with :98-112)."""
    import ra_train as rt
    h = [None] * nlayers
    prev = x
    stats = run_cnn.batch_stats = {}
    for ii in range(nlayers):
      if pool[ii] > 2:
        raise RecAttendError('cnn: pool ratio %d not fused (path uses 1 or 2)' % pool[ii])
      meta = dict(transposed=False, stride=1, pool=pool[ii] if pool[ii] > 1 else 1, relu=act[ii] is not None,
                  chan_map=None)
      bn = _bn(ii, cp) if use_bn[ii] else None
      y, mean, var = rt.ConvBNActPool.apply(rt._pad_channels(prev), w[ii], b[ii], bn[1] if bn else None,
                                            bn[0] if bn else None, meta)
      if bn:
        _ema_update(bn[2], bn[3], mean, var)
        stats['{}_{}_{}'.format(scope, ii, cp)] = (mean, var)
      h[ii] = prev = y
    return h

  def run_cnn(x, copy_idx=None):
    cp = copy[0] if copy_idx is None else copy_idx
    for ii in range(nlayers):
      if act[ii] is not None and act[ii] is not relu_fn and getattr(act[ii], '__name__',
                                                                     '') != 'relu':
        raise RecAttendError('cnn: only relu / None activations are fused')
    if _is_train(phase_train):
      h = run_cnn_train(x, cp)
      if copy_idx is None:
        copy[0] += 1
      return h
    h = [None] * nlayers
    prev = x
    for ii in range(nlayers):
      cin = prev.shape[3]
      if cin % 4:
        prev = torch.nn.functional.pad(prev, (0, 4 - cin % 4))
      wp, sc, sh = layer_params(ii, cp, prev, cin_kernel=prev.shape[3])
      h[ii] = ops.conv3x3(prev.contiguous(), wp, sc, sh, ch[ii + 1], relu=act[ii] is not None,
                          pool=pool[ii] if pool[ii] > 1 else 1)
      if pool[ii] > 2:
        raise RecAttendError('cnn: pool ratio %d not fused (path uses 1 or 2)' % pool[ii])
      prev = h[ii]
    if copy_idx is None:
      copy[0] += 1
    return h

  run_cnn.local = {}
  run_cnn.reset_copy = lambda: copy.__setitem__(0, 0)
  run_cnn.declare_copies = declare_copies
  run_cnn.layer_params = layer_params
  run_cnn.w, run_cnn.b = w, b
  return run_cnn


def dcnn(f, ch, pool, act, use_bn, skip_ch=None, phase_train=None, wd=None, scope='dcnn',
         model=None, init_weights=None, frozen=None):
  """nnlib.py:260-404.  Returns run_dcnn(x, skip=None) -> list of the N layer outputs.

  Layer = [concat(prev, skip[i])] -> conv2d_transpose(w[f,f,out,in], stride pool[i], SAME) + b
  -> BN(copy) -> ReLU, as ONE fused kernel (the concat is two source pointers, the stride-2
  transpose is a conv over the zero-stuffed input with flipped taps)."""
  nlayers = len(f)
  w = [None] * nlayers
  b = [None] * nlayers
  in_chs = [None] * nlayers
  in_ch = ch[0]
  for ii in range(nlayers):
    if f[ii] != 3:
      raise RecAttendError('dcnn: filter size %d not built' % f[ii])
    out_ch = ch[ii + 1]
    if skip_ch is not None and skip_ch[ii] is not None:
      in_ch += skip_ch[ii]
    iw = init_weights[ii] if init_weights is not None and init_weights[ii] is not None else None
    trainable = not (frozen is not None and frozen[ii])
    w[ii] = weight_variable([f[ii], f[ii], out_ch, in_ch], name='w',
                            init_val=None if iw is None else iw['w'], wd=wd,
                            trainable=trainable)
    b[ii] = weight_variable([out_ch], init_val=None if iw is None else iw['b'], name='b',
                            trainable=trainable)
    in_chs[ii] = in_ch
    in_ch = out_ch
    if model is not None:
      model['{}_w_{}'.format(scope, ii)] = w[ii]
      model['{}_b_{}'.format(scope, ii)] = b[ii]
  copy = [0]
  cache = _LayerCache()

  def _bn(ii, cp):
    iw = init_weights[ii] if init_weights is not None and init_weights[ii] is not None else None
    ib = None if iw is None else iw.get('beta_{}'.format(cp))
    ig = None if iw is None else iw.get('gamma_{}'.format(cp))
    return _bn_register(model if model is not None else run_dcnn.local, '{}_{}_{}'.format(
        scope, ii, cp), ch[ii + 1], ib, ig)

  def declare_copies(n):
    for ii in range(nlayers):
      if use_bn[ii]:
        for cp in range(n):
          _bn(ii, cp)

  def layer_params(ii, cp, x_like, c_prev, c_skip):
    """c_prev / c_skip: kernel channel counts (each a multiple of 4) of the two sources; the
    filter's input channels are [prev (ch) | skip (skip_ch)] (nnlib.py:365)."""
    n_prev = in_chs[ii] - (skip_ch[ii] if skip_ch is not None and skip_ch[ii] else 0)
    n_skip = in_chs[ii] - n_prev
    cmap = [k if k < n_prev else -1 for k in range(c_prev)] + \
        [n_prev + k if k < n_skip else -1 for k in range(c_skip)]
    wp = cache.get(('w', ii, c_prev, c_skip), [w[ii]], lambda: _dev(
        ops.pack_conv_weights(w[ii], cin_kernel=c_prev + c_skip, chan_map=cmap,
                              transposed=True), x_like))
    bn = _bn(ii, cp) if use_bn[ii] else None
    srcs = [b[ii]] + (list(bn) if bn else [])
    sc, sh = cache.get(('bn', ii, cp), srcs, lambda: tuple(
        _dev(a, x_like) for a in ops.fold_bn(b[ii], ch[ii + 1], bn)))
    return wp, sc, sh

  def run_dcnn_train(x, skip, cp):
    """phase_train = True: [concat(prev, skip)] -> conv2d_transpose + b -> BN on the batch moments -> ReLU, one autograd
    node per layer on the training step's kernels (nnlib.py:362-400 with :98-112).  The concat is ONE packed kernel
    input whose chan_map sends every packed channel to its row of the [3,3,out,in] filter."""
    import ra_train as rt
    h = [None] * nlayers
    prev = x
    stats = run_dcnn.batch_stats = {}
    for ii in range(nlayers):
      if pool[ii] not in (1, 2):
        raise RecAttendError('dcnn: unpool ratio %d not built' % pool[ii])
      sk = skip[ii] if skip is not None else None
      cmap = None
      if sk is not None:
        n_prev, n_skip = prev.shape[3], sk.shape[3]
        xp, skp = rt._pad_channels(prev), rt._pad_channels(sk)
        cmap = list(range(n_prev)) + [-1] * (xp.shape[3] - n_prev) + [n_prev + k for k in range(n_skip)] + \
            [-1] * (skp.shape[3] - n_skip)
        if cmap == list(range(len(cmap))):
          cmap = None
        prev = torch.cat([xp, skp], dim=3)
      meta = dict(transposed=True, stride=pool[ii], pool=1, relu=act[ii] is not None, chan_map=cmap)
      bn = _bn(ii, cp) if use_bn[ii] else None
      y, mean, var = rt.ConvBNActPool.apply(rt._pad_channels(prev), w[ii], b[ii], bn[1] if bn else None,
                                            bn[0] if bn else None, meta)
      if bn:
        _ema_update(bn[2], bn[3], mean, var)
        stats['{}_{}_{}'.format(scope, ii, cp)] = (mean, var)
      h[ii] = prev = y
    return h

  def run_dcnn(x, skip=None, copy_idx=None):
    cp = copy[0] if copy_idx is None else copy_idx
    if _is_train(phase_train):
      h = run_dcnn_train(x, skip, cp)
      if copy_idx is None:
        copy[0] += 1
      return h
    h = [None] * nlayers
    prev = x
    for ii in range(nlayers):
      sk = skip[ii] if skip is not None else None
      if prev.shape[3] % 4:
        prev = torch.nn.functional.pad(prev, (0, 4 - prev.shape[3] % 4))
      if sk is not None and sk.shape[3] % 4:
        sk = torch.nn.functional.pad(sk, (0, 4 - sk.shape[3] % 4))
      if pool[ii] not in (1, 2):
        raise RecAttendError('dcnn: unpool ratio %d not built' % pool[ii])
      wp, sc, sh = layer_params(ii, cp, prev, prev.shape[3], 0 if sk is None else sk.shape[3])
      h[ii] = ops.conv3x3(prev.contiguous(), wp, sc, sh, ch[ii + 1], relu=act[ii] is not None,
                          pool=1, src1=None if sk is None else sk.contiguous(),
                          upsample=(pool[ii] == 2))
      prev = h[ii]
    if copy_idx is None:
      copy[0] += 1
    return h

  run_dcnn.local = {}
  run_dcnn.reset_copy = lambda: copy.__setitem__(0, 0)
  run_dcnn.declare_copies = declare_copies
  run_dcnn.layer_params = layer_params
  run_dcnn.w, run_dcnn.b, run_dcnn.in_chs = w, b, in_chs
  return run_dcnn


def dropout(x, keep_prob, phase_train):
  """nnlib.py:405-409: tf.nn.dropout(x, keep) with keep = 1 at eval (identity) and keep_prob in training (kept values
  scaled by 1 / keep_prob; the mask is torch's generator's draw — the reference's is TF's, neither is reproducible from
  the other)."""
  if not _is_train(phase_train) or keep_prob is None or float(keep_prob) >= 1.0:
    return x
  return torch.nn.functional.dropout(x, p=1.0 - float(keep_prob), training=True)


_ACT_CODE = {None: None, 'relu': 'relu', 'sigmoid': 'sigmoid', 'softmax': 'softmax',
             'tanh': 'tanh'}


def _act_name(fn):
  if fn is None:
    return None
  if isinstance(fn, str):
    return _ACT_CODE[fn]
  name = getattr(fn, '__name__', '')
  if name in _ACT_CODE:
    return name
  raise RecAttendError('mlp: activation %r not built (relu/sigmoid/softmax/tanh/None)' % fn)


def sigmoid(x):
  raise RecAttendError('nnlib.sigmoid is an activation tag for mlp(); it is fused, not callable')


def softmax(x):
  raise RecAttendError('nnlib.softmax is an activation tag for mlp(); it is fused, not callable')


def tanh(x):
  raise RecAttendError('nnlib.tanh is an activation tag for mlp(); it is fused, not callable')


def mlp(dims, act, add_bias=True, dropout_keep=None, phase_train=None, wd=None, scope='mlp',
        model=None, init_weights=None, frozen=None):
  """nnlib.py:414-495.  Returns run_mlp(x) -> list of layer outputs; act entries are
  nnlib.relu / nnlib.sigmoid / nnlib.softmax / nnlib.tanh / None (or their names)."""
  nlayers = len(dims) - 1
  w = [None] * nlayers
  b = [None] * nlayers
  for ii in range(nlayers):
    iw = init_weights[ii] if init_weights is not None and init_weights[ii] is not None else None
    trainable = not (frozen is not None and frozen[ii])
    w[ii] = weight_variable([dims[ii], dims[ii + 1]], init_val=None if iw is None else iw['w'],
                            wd=wd, name='w', trainable=trainable)
    if add_bias:
      b[ii] = weight_variable([dims[ii + 1]], init_val=None if iw is None else iw['b'],
                              name='b', trainable=trainable)
    if model is not None:
      model['{}_w_{}'.format(scope, ii)] = w[ii]
      if add_bias:
        model['{}_b_{}'.format(scope, ii)] = b[ii]

  def run_mlp_train(x, x1=None):
    """The differentiable form (training graphs): dropout (nnlib.py:484-486), x W + b as one addmm per layer — the
    library GEMM the training step also uses for [B, <= 1408] x [<= 1408, <= 256] products — and the activation
    (nnlib.py:487-491)."""
    h = [None] * nlayers
    prev = x if x1 is None else torch.cat([x, x1], dim=1)
    for ii in range(nlayers):
      if dropout_keep is not None and dropout_keep[ii] is not None:
        prev = dropout(prev, dropout_keep[ii], phase_train)
      out = torch.addmm(b[ii], prev, w[ii]) if b[ii] is not None else prev @ w[ii]
      a = _act_name(act[ii])
      if a == 'relu':
        out = torch.relu(out)
      elif a == 'sigmoid':
        out = torch.sigmoid(out)
      elif a == 'tanh':
        out = torch.tanh(out)
      elif a == 'softmax':
        out = torch.softmax(out, dim=1)
      h[ii] = prev = out
    return h

  def run_mlp(x, x1=None):
    """x1: optional second input whose columns follow x's (fused concat)."""
    if _is_train(phase_train) or (torch.is_grad_enabled() and (x.requires_grad or any(
        t is not None and t.requires_grad for t in w + b))):
      return run_mlp_train(x, x1)
    h = [None] * nlayers
    prev, extra = x.contiguous(), x1
    for ii in range(nlayers):
      out = torch.empty((prev.shape[0], dims[ii + 1]), dtype=torch.float32, device=prev.device)
      ops.dense(prev, w[ii], b[ii], _act_name(act[ii]), out, dims[ii + 1],
                x1=None if extra is None else extra.contiguous())
      h[ii] = out
      prev, extra = out, None
    return h

  run_mlp.w, run_mlp.b = w, b
  return run_mlp


def lstm(inp_dim, hid_dim, wd=None, scope='lstm', model=None, init_weights=None, frozen=False):
  """nnlib.py:498-651.  Returns unroll(inp, state) -> (state, g_i, g_f, g_o), state = [c | h].

  The fused controller kernel (ra_controller_f32) runs this cell inside the glimpse loop;
  this stand-alone closure composes the same math from the dense kernel for API parity."""
  names = ['w_xi', 'w_hi', 'b_i', 'w_xf', 'w_hf', 'b_f', 'w_xu', 'w_hu', 'b_u', 'w_xo', 'w_ho',
           'b_o']
  if init_weights is None:
    init_weights = {n: None for n in names}
  trainable = not frozen
  P = {}
  for n in names:
    if n.startswith('w_x'):
      shape, init = [inp_dim, hid_dim], None
    elif n.startswith('w_h'):
      shape, init = [hid_dim, hid_dim], None
    else:
      shape = [hid_dim]
      init = constant_initializer(1.0 if n == 'b_f' else 0.0)  # nnlib.py:547,567,588,608
    P[n] = weight_variable(shape, init_val=init_weights[n], initializer=init,
                           wd=wd if n[0] == 'w' else None, name=n, trainable=trainable)
    if model is not None:
      model['{}_{}'.format(scope, n)] = P[n]

  def unroll_train(inp, state):
    """The differentiable cell (a training graph): the four gates as ONE addmm on [w_x; w_h] in gate order (i, f, o, u),
    the pointwise half and its adjoint on ra_lstm_cell_f32 / _bwd (ra_train.LSTMCell)."""
    import ra_train as rt
    c, h = state[:, :hid_dim], state[:, hid_dim:]
    wcat = torch.cat([torch.cat([P['w_x' + g], P['w_h' + g]], dim=0) for g in 'ifou'], dim=1)
    bcat = torch.cat([P['b_' + g] for g in 'ifou'])
    pre = torch.addmm(bcat, torch.cat([inp, h], dim=1), wcat)
    h2, c2 = rt.LSTMCell.apply(pre, c)
    g = torch.sigmoid(pre[:, :3 * hid_dim])
    return torch.cat([c2, h2], dim=1), g[:, :hid_dim], g[:, hid_dim:2 * hid_dim], g[:, 2 * hid_dim:]

  def unroll(inp, state):
    if torch.is_grad_enabled() and (inp.requires_grad or state.requires_grad or any(
        t.requires_grad for t in P.values())):
      return unroll_train(inp, state)
    c = state[:, :hid_dim].contiguous()
    h = state[:, hid_dim:].contiguous()
    B = inp.shape[0]

    def gate(g, a):
      wcat = torch.cat([P['w_x' + g], P['w_h' + g]], dim=0).contiguous()
      out = torch.empty((B, hid_dim), dtype=torch.float32, device=inp.device)
      ops.dense(inp.contiguous(), wcat, P['b_' + g], a, out, hid_dim, x1=h)
      return out

    g_i, g_f, g_o, u = gate('i', 'sigmoid'), gate('f', 'sigmoid'), gate('o', 'sigmoid'), gate(
        'u', 'tanh')
    c = g_f * c + g_i * u
    h = g_o * torch.tanh(c)
    return torch.cat([c, h], dim=1), g_i, g_f, g_o

  unroll.params = P
  return unroll
