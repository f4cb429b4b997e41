"""Thin torch-tensor wrappers over the C ABI (include/recattend.h).

Every function here launches hand-written HIP kernels from librecattend.so on the current
torch stream.  Inputs must be float32 CUDA tensors (contiguous); there is no CPU path —
calling with CPU tensors raises.  Host-side weight repacking (ra_conv_pack_weights,
ra_conv_fold_bn, ra_ctrl_pack_weights) runs in the library's C++ and works without a GPU.
"""
import ctypes as C

import numpy as np
import torch

import ra_native as rn
from ra_native import check, ptr

BN_EPS = 1e-3  # nnlib.py:119


def _need_cuda(*ts):
  for t in ts:
    if t is not None and not t.is_cuda:
      raise rn.RecAttendError('recattend kernels need CUDA (HIP) tensors; got a CPU tensor. '
                              'There is no CPU fallback.')
    if t is not None and (t.dtype != torch.float32 or not t.is_contiguous()):
      raise rn.RecAttendError('recattend kernels need contiguous float32 tensors')


def _np32(a):
  if isinstance(a, torch.Tensor):
    a = a.detach().cpu().numpy()
  return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------------------- host packing


def cout_padded(cout):
  return rn.lib().ra_conv_cout_padded(int(cout))


def pack_conv_weights(w, cin_kernel=None, chan_map=None, transposed=False):
  """TF-layout filter -> packed B-operand order (numpy, host).  w [3,3,Ci,Co] or, when
  transposed, the conv2d_transpose filter [3,3,Co,Ci]."""
  w = _np32(w)
  if w.shape[0] != 3 or w.shape[1] != 3:
    raise rn.RecAttendError('only 3x3 filters are supported, got %r' % (w.shape,))
  cin_w, cout = (w.shape[3], w.shape[2]) if transposed else (w.shape[2], w.shape[3])
  cin = cin_w if cin_kernel is None else int(cin_kernel)
  if chan_map is None and cin != cin_w:
    chan_map = list(range(cin_w)) + [-1] * (cin - cin_w)
  n = rn.lib().ra_conv_packed_floats(cin, cout)
  if n == 0:
    raise rn.RecAttendError('unsupported conv shape Cin=%d Cout=%d' % (cin, cout))
  out = np.empty(n, dtype=np.float32)
  cm = None
  if chan_map is not None:
    cm = np.ascontiguousarray(chan_map, dtype=np.int32)
    assert cm.shape[0] == cin
  check(rn.lib().ra_conv_pack_weights(ptr(w), cin_w, cout, cin, ptr(cm),
                                      rn.RA_CONV_TRANSPOSED if transposed else 0, ptr(out)),
        'ra_conv_pack_weights')
  return out


def fold_bn(bias, cout, bn=None):
  """(scale, shift) [CoutP] numpy.  bn = (beta, gamma, ema_mean, ema_var) or None."""
  cp = cout_padded(cout)
  scale = np.empty(cp, dtype=np.float32)
  shift = np.empty(cp, dtype=np.float32)
  b = None if bias is None else _np32(bias)
  if bn is None:
    args = (None, None, None, None)
  else:
    args = tuple(_np32(a) for a in bn)
  keep = (b,) + args
  check(rn.lib().ra_conv_fold_bn(ptr(b), ptr(args[0]), ptr(args[1]), ptr(args[2]), ptr(args[3]),
                                 int(cout), C.c_float(BN_EPS), ptr(scale), ptr(shift)),
        'ra_conv_fold_bn')
  del keep
  return scale, shift


def make_ctrl_desc(G, Cf, hid, iters, n_gmlp, n_cmlp, mlp_dim, H, W, Fh, Fw, squash, fixed_var,
                   dynamic_var, fixed_gamma):
  return rn.CtrlDesc(int(G), int(Cf), int(hid), int(iters), int(n_gmlp), int(n_cmlp),
                     int(mlp_dim), int(H), int(W), int(Fh), int(Fw), int(bool(squash)),
                     int(bool(fixed_var)), int(bool(dynamic_var)), int(bool(fixed_gamma)))


def pack_ctrl_weights(desc, lstm, gmlp, cmlp):
  """lstm: dict with keys w_xi,w_hi,b_i,...; gmlp/cmlp: lists [(w,b),...].  -> numpy."""
  n = rn.lib().ra_ctrl_packed_floats(C.byref(desc))
  if n == 0:
    raise rn.RecAttendError('unsupported controller descriptor')
  order = ['w_xi', 'w_hi', 'b_i', 'w_xf', 'w_hf', 'b_f', 'w_xu', 'w_hu', 'b_u', 'w_xo', 'w_ho',
           'b_o']
  la = [_np32(lstm[k]) for k in order]
  ga = [_np32(a) for wb in gmlp for a in wb]
  ca = [_np32(a) for wb in cmlp for a in wb]
  arr = lambda xs: (C.c_void_p * len(xs))(*[x.ctypes.data for x in xs])
  out = np.empty(n, dtype=np.float32)
  check(rn.lib().ra_ctrl_pack_weights(C.byref(desc), arr(la), arr(ga), arr(ca), ptr(out)),
        'ra_ctrl_pack_weights')
  return out


def ctrl_split_supported(desc):
  return bool(rn.lib().ra_ctrl_split_supported(C.byref(desc)))


def pack_ctrl_split_weights(desc, lstm, gmlp, cmlp):
  n = rn.lib().ra_ctrl_split_packed_floats(C.byref(desc))
  if n == 0:
    raise rn.RecAttendError('descriptor not supported by the split controller')
  order = ['w_xi', 'w_hi', 'b_i', 'w_xf', 'w_hf', 'b_f', 'w_xu', 'w_hu', 'b_u', 'w_xo', 'w_ho',
           'b_o']
  la = [_np32(lstm[k]) for k in order]
  ga = [_np32(a) for wb in gmlp for a in wb]
  ca = [_np32(a) for wb in cmlp for a in wb]
  arr = lambda xs: (C.c_void_p * len(xs))(*[x.ctypes.data for x in xs])
  out = np.empty(n, dtype=np.float32)
  check(rn.lib().ra_ctrl_split_pack_weights(C.byref(desc), arr(la), arr(ga), arr(ca), ptr(out)),
        'ra_ctrl_split_pack_weights')
  return out


_CUS = []


def cu_count():
  """Compute units of the current device (256 on MI355X)."""
  if not _CUS:
    _CUS.append(int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count))
  return _CUS[0]


def ctrl_split_workspace(desc, B, device):
  """Zero-filled exchange workspace (+ status word) for ONE stream of launches."""
  nb = rn.lib().ra_ctrl_split_workspace_bytes(C.byref(desc), B)
  return (torch.zeros((nb + 7) // 8, dtype=torch.int64, device=device),
          torch.zeros(1, dtype=torch.int32, device=device))


def controller_split(desc, feat, wp, h_last, ctrl_out, gmaps, attn, ws, status):
  _need_cuda(feat, wp, h_last, ctrl_out, gmaps, attn)
  check(rn.lib().ra_controller_split_f32(C.byref(desc), ptr(feat), ptr(wp), feat.shape[0],
                                         ptr(h_last), ptr(ctrl_out), ptr(gmaps), ptr(attn),
                                         ptr(ws), ws.numel() * 8, ptr(status), rn.stream_ptr()),
        'ra_controller_split_f32')


def ctrl_batch_group(desc, B):
  """Images that share one set of 16 controller workgroups in a controller_batch (K2b) launch of B images."""
  return int(rn.lib().ra_ctrl_batch_group_images(C.byref(desc), int(B)))


def ctrl_batch_supported(desc):
  return bool(rn.lib().ra_ctrl_batch_supported(C.byref(desc)))


def ctrl_batch_workspace(desc, B, device):
  """Zero-filled exchange workspace (+ status word) of the group-shared controller, for ONE stream of launches."""
  nb = rn.lib().ra_ctrl_batch_workspace_bytes(C.byref(desc), B)
  return (torch.zeros((nb + 7) // 8, dtype=torch.int64, device=device),
          torch.zeros(1, dtype=torch.int32, device=device))


def controller_batch(desc, feat, wp, h_last, ctrl_out, gmaps, attn, ws, status, xcd_offset=-1):
  """K2b: ra_controller_split_f32's recurrence with the weight slices shared by groups of ctrl_batch_group(desc, B) images.
  xcd_offset >= 0: the XCD-local exchange, group g on XCD (g + xcd_offset) % 8 (ra_controller_batch_xcd_f32: the caller keeps
  concurrent launches on different XCDs); -1: agent-scope atomics."""
  _need_cuda(feat, wp, h_last, ctrl_out, gmaps, attn)
  check(rn.lib().ra_controller_batch_xcd_f32(C.byref(desc), ptr(feat), ptr(wp), feat.shape[0], ptr(h_last), ptr(ctrl_out),
                                             ptr(gmaps), ptr(attn), ptr(ws), ws.numel() * 8, ptr(status), int(xcd_offset), rn.stream_ptr()),
        'ra_controller_batch_xcd_f32')


# ---------------------------------------------------------------------------- device ops


def conv3x3(src0, wp, scale, shift, cout, relu=True, pool=1, src1=None, upsample=False,
            out=None, plane=None, plane_chan=-1, bf16=False):
  """One fused conv layer.  src0 [B,Hs,Ws,C0] (+ src1 [B,Hs,Ws,C1]) -> [B,Ho,Wo,cout].
  bf16: bf16 operands on the bf16 MFMA, float32 accumulation (the training step's mixed-precision mode)."""
  _need_cuda(src0, src1, wp, scale, shift, out)
  B, Hs, Ws, C0 = src0.shape
  C1 = 0 if src1 is None else src1.shape[3]
  up = 2 if upsample else 1
  Ho, Wo = Hs * up // pool, Ws * up // pool
  if out is None:
    out = torch.empty((B, Ho, Wo, cout), dtype=torch.float32, device=src0.device)
  fn = rn.lib().ra_conv3x3_bf16ops_f32 if bf16 else rn.lib().ra_conv3x3_f32
  check(fn(ptr(src0), C0, ptr(src1), C1, B, Hs, Ws, int(upsample), ptr(wp), ptr(scale), ptr(shift), int(cout),
           int(relu), int(pool), ptr(plane), int(plane_chan), ptr(out), rn.stream_ptr()),
        'ra_conv3x3_bf16ops_f32' if bf16 else 'ra_conv3x3_f32')
  return out


def conv_wino_supported(cin, cout, pool, H, W):
  return bool(rn.lib().ra_conv_wino_supported(int(cin), int(cout), int(pool), int(H), int(W)))


def pack_wino_weights(w):
  """TF-layout [3,3,Cin,Cout] filter -> transformed filters G g G^T in MFMA B-operand order (numpy, host)."""
  w = _np32(w)
  cin, cout = w.shape[2], w.shape[3]
  n = rn.lib().ra_conv_wino_packed_floats(cin, cout)
  if w.shape[0] != 3 or w.shape[1] != 3 or n == 0:
    raise rn.RecAttendError('unsupported Winograd conv shape %r' % (w.shape,))
  out = np.empty(n, dtype=np.float32)
  check(rn.lib().ra_conv_wino_pack_weights(ptr(w), cin, cout, ptr(out)), 'ra_conv_wino_pack_weights')
  return out


def conv_split_supported(cin, cout, pool, H, W):
  return bool(rn.lib().ra_conv_split_supported(int(cin), int(cout), int(pool), int(H), int(W)))


def pack_split_weights(w, cin_kernel=None, chan_map=None):
  """TF-layout [3,3,Cin,Cout] filter -> its exact three-piece bf16 split in K1s's B-operand order (numpy, host; bf16 pairs in float32 words).
  cin_kernel / chan_map (as pack_conv_weights): the kernel's input has cin_kernel channels, channel c of it is the filter's input
  channel chan_map[c] (-1: a channel the filter does not read — zero rows)."""
  w = _np32(w)
  if cin_kernel is not None or chan_map is not None:
    ck = int(cin_kernel if cin_kernel is not None else len(chan_map))
    cm = list(chan_map) if chan_map is not None else list(range(w.shape[2])) + [-1] * (ck - w.shape[2])
    wk = np.zeros((3, 3, ck, w.shape[3]), np.float32)
    for c, m_ in enumerate(cm[:ck]):
      if m_ >= 0:
        wk[:, :, c, :] = w[:, :, m_, :]
    w = wk
  cin, cout = w.shape[2], w.shape[3]
  n = rn.lib().ra_conv_split_packed_halfs(cin, cout)
  if w.shape[0] != 3 or w.shape[1] != 3 or n == 0:
    raise rn.RecAttendError('unsupported split-precision conv shape %r' % (w.shape,))
  out = np.empty(n, dtype=np.int16)
  check(rn.lib().ra_conv_split_pack_weights(ptr(w), cin, cout, ptr(out)), 'ra_conv_split_pack_weights')
  return out.view(np.float32)  # carried as a float32 tensor like every other packed filter (two bf16 pieces per word)


def pack_split_weights_dev(w, cin, cout, transposed=False):
  """The same packing from a DEVICE filter (the training step: the weights change every step).  cin / cout: the conv's input
  / output channels; transposed: w is laid out [3,3,cout,cin] and the taps flip (a conv2d_transpose filter, or the filter of
  a cnn layer whose data gradient this conv is)."""
  _need_cuda(w)
  n = rn.lib().ra_conv_split_packed_halfs(int(cin), int(cout))
  if n == 0 or w.numel() != 9 * cin * cout:
    raise rn.RecAttendError('unsupported split-precision conv shape Cin=%d Cout=%d' % (cin, cout))
  out = torch.empty(n // 2, dtype=torch.float32, device=w.device)
  check(rn.lib().ra_conv_split_pack_weights_dev(ptr(w.contiguous()), int(cin), int(cout), int(bool(transposed)), ptr(out),
                                                rn.stream_ptr()), 'ra_conv_split_pack_weights_dev')
  return out


def conv_split(x, wp, scale, shift, cout, relu=True, pool=1, out=None, plane=None, plane_chan=-1):
  """conv3x3 SAME + folded BN + ReLU + pool as a direct convolution on the bf16 matrix pipe at float32 accuracy
  (ra_conv_split_f32: three bf16 pieces per operand, six piece products).  x [B,H,W,Cin]; plane [B,H,W]: stands in for channel
  plane_chan of x (ra_conv_split_plane_f32)."""
  _need_cuda(x, wp, scale, shift, out, plane)
  B, H, W, cin = x.shape
  if out is None:
    out = torch.empty((B, H // pool, W // pool, cout), dtype=torch.float32, device=x.device)
  if plane is not None:
    check(rn.lib().ra_conv_split_plane_f32(ptr(x), B, H, W, cin, ptr(plane), int(plane_chan), ptr(wp), ptr(scale), ptr(shift), int(cout),
                                           int(relu), int(pool), ptr(out), rn.stream_ptr()), 'ra_conv_split_plane_f32')
    return out
  check(rn.lib().ra_conv_split_f32(ptr(x), B, H, W, cin, ptr(wp), ptr(scale), ptr(shift), int(cout), int(relu), int(pool),
                                   ptr(out), rn.stream_ptr()), 'ra_conv_split_f32')
  return out


def conv_wino(x, wp, scale, shift, cout, relu=True, pool=1, out=None):
  """conv3x3 SAME + folded BN + ReLU + pool as Winograd F(2x2,3x3) (ra_conv_wino_f32).  x [B,H,W,Cin]."""
  _need_cuda(x, wp, scale, shift, out)
  B, H, W, cin = x.shape
  if out is None:
    out = torch.empty((B, H // pool, W // pool, cout), dtype=torch.float32, device=x.device)
  check(rn.lib().ra_conv_wino_f32(ptr(x), B, H, W, cin, ptr(wp), ptr(scale), ptr(shift), int(cout), int(relu), int(pool),
                                  ptr(out), rn.stream_ptr()), 'ra_conv_wino_f32')
  return out


def conv_pair_wino_supported(cin, cout_a, cout_b, pool_b, H, W):
  return bool(rn.lib().ra_conv_pair_wino_supported(int(cin), int(cout_a), int(cout_b), int(pool_b), int(H), int(W)))


def conv_pair_wino(x, wpA, scA, shA, wpB_wino, scB, shB, reluA=True, reluB=True, out=None):
  """The fused pair 8 -> 16 -> 16, pool 2, with layer B as Winograd (ra_conv_pair_wino_f32).  x [B,H,W,8]."""
  _need_cuda(x, wpA, scA, shA, wpB_wino, scB, shB, out)
  B, H, W, _ = x.shape
  if out is None:
    out = torch.empty((B, H // 2, W // 2, 16), dtype=torch.float32, device=x.device)
  check(rn.lib().ra_conv_pair_wino_f32(ptr(x), B, H, W, ptr(wpA), ptr(scA), ptr(shA), int(reluA), ptr(wpB_wino), ptr(scB),
                                       ptr(shB), int(reluB), ptr(out), rn.stream_ptr()), 'ra_conv_pair_wino_f32')
  return out


def poison_lds():
  """Test aid: leave NaN in every CU's LDS (see ra_debug_poison_lds)."""
  check(rn.lib().ra_debug_poison_lds(rn.stream_ptr()), 'ra_debug_poison_lds')


def park_xcd(xcd, n_wg, lds_bytes=100 * 1024, millis=100, resident=None, stream=None):
  """Test aid: hold n_wg CUs' worth of LDS on one XCD for `millis` ms (ra_debug_park_xcd); resident: int32 device counter."""
  check(rn.lib().ra_debug_park_xcd(int(xcd), int(n_wg), int(lds_bytes), int(millis), ptr(resident) if resident is not None else None,
                                   rn.stream_ptr() if stream is None else stream.cuda_stream), 'ra_debug_park_xcd')


def conv_pair_supported(cin, cout_a, cout_b):
  return bool(rn.lib().ra_conv_pair_supported(int(cin), int(cout_a), int(cout_b)))


def conv_pair(src, wpA, scA, shA, coutA, wpB, scB, shB, coutB, poolB=1, upsampleA=False,
              reluA=True, reluB=True, out=None, plane=None, plane_chan=-1):
  """Two fused conv layers (A: no pool / optional stride-2 transposed; B: pool 1|2)."""
  _need_cuda(src, wpA, scA, shA, wpB, scB, shB, out)
  B, Hs, Ws, C0 = src.shape
  up = 2 if upsampleA else 1
  Ho, Wo = Hs * up // poolB, Ws * up // poolB
  if out is None:
    out = torch.empty((B, Ho, Wo, coutB), dtype=torch.float32, device=src.device)
  check(rn.lib().ra_conv_pair_f32(ptr(src), C0, B, Hs, Ws, int(upsampleA), ptr(wpA), ptr(scA),
                                  ptr(shA), int(coutA), int(reluA), ptr(wpB), ptr(scB), ptr(shB),
                                  int(coutB), int(reluB), int(poolB), ptr(plane), int(plane_chan),
                                  ptr(out), rn.stream_ptr()),
        'ra_conv_pair_f32')
  return out


def first_cache_supported(cin, cout_a, cout_b, pool_b, H, W):
  return bool(rn.lib().ra_conv_first_cache_supported(int(cin), int(cout_a), int(cout_b), int(pool_b), int(H), int(W)))


def first_cache_alloc(B, H, W, device):
  return torch.zeros((rn.lib().ra_conv_first_cache_floats(B, H, W),), dtype=torch.float32, device=device)


def first_cache(img, wpA, cout_a, plane_chan, cache):
  """Once per forward: the image channels' share of the first conv layer (ra_conv_first_cache_f32)."""
  _need_cuda(img, wpA, cache)
  B, H, W, C = img.shape
  assert C == 4
  check(rn.lib().ra_conv_first_cache_f32(ptr(img), B, H, W, ptr(wpA), int(cout_a), int(plane_chan), ptr(cache),
                                         rn.stream_ptr()), 'ra_conv_first_cache_f32')


def fill_rider_ok(t):
  """Can `t` be filled by the rider of conv_pair_fill_cache (contiguous, 16-byte aligned, % 4, < 2 GiB)?"""
  return t.is_contiguous() and t.data_ptr() % 16 == 0 and t.numel() % 4 == 0 and t.numel() * 4 < (1 << 31)


def conv_pair_fill_cache(img, plane, plane_chan, wpA, scA, shA, wpB, scB, shB, coutB, cache, out, reluA=True, reluB=True,
                         fill=None, fill_value=0.0):
  """First timestep (canvas plane all zero): the plain first pair, writing the cache on the way.
  fill: a tensor set to fill_value by the same launch (the decode loop's y_out prefill)."""
  _need_cuda(img, plane, wpA, scA, shA, wpB, scB, shB, cache, out, fill)
  B, H, W = plane.shape
  if fill is not None and not fill_rider_ok(fill):
    raise rn.RecAttendError('conv_pair_fill_cache: the rider fill needs a contiguous, 16-byte aligned tensor of 4k floats < 2 GiB')
  check(rn.lib().ra_conv_pair_fill_cache_rider_f32(ptr(img), ptr(plane), int(plane_chan), B, H, W, ptr(wpA), ptr(scA),
                                                   ptr(shA), int(reluA), ptr(wpB), ptr(scB), ptr(shB), int(coutB),
                                                   int(reluB), ptr(cache), ptr(out), ptr(fill),
                                                   0 if fill is None else fill.numel(), C.c_float(fill_value),
                                                   rn.stream_ptr()),
        'ra_conv_pair_fill_cache_rider_f32')
  return out


def conv_pair_cached(cache, plane, plane_chan, wpA, scA, shA, wpB, scB, shB, coutB, out, reluA=True, reluB=True):
  """Per timestep: the first controller-CNN pair from the cached image part + the canvas plane."""
  _need_cuda(cache, plane, wpA, scA, shA, wpB, scB, shB, out)
  B, H, W = plane.shape
  check(rn.lib().ra_conv_pair_cached_f32(ptr(cache), ptr(plane), int(plane_chan), B, H, W, ptr(wpA), ptr(scA),
                                         ptr(shA), int(reluA), ptr(wpB), ptr(scB), ptr(shB), int(coutB),
                                         int(reluB), ptr(out), rn.stream_ptr()), 'ra_conv_pair_cached_f32')
  return out


def controller(desc, feat, wp, h_last, ctrl_out, gmaps, attn):
  _need_cuda(feat, wp, h_last, ctrl_out, gmaps, attn)
  B = feat.shape[0]
  check(rn.lib().ra_controller_f32(C.byref(desc), ptr(feat), ptr(wp), B, ptr(h_last),
                                   ptr(ctrl_out), ptr(gmaps), ptr(attn), rn.stream_ptr()),
        'ra_controller_f32')


def extract_direct(img, chan0, attn, Fh, Fw, Cp, use_gamma, patch, canvas=None, canvas_chan=-1):
  _need_cuda(img, attn, patch, canvas)
  B, H, W, Ci = img.shape
  check(rn.lib().ra_extract_direct_f32(ptr(img), Ci, chan0, ptr(canvas), int(canvas_chan), ptr(attn),
                                       B, H, W, Fh, Fw, Cp, int(use_gamma), ptr(patch),
                                       rn.stream_ptr()), 'ra_extract_direct_f32')


def extract_conv0_supported(Cp, Fh, Fw, cout, pool):
  return bool(rn.lib().ra_extract_conv0_supported(int(Cp), int(Fh), int(Fw), int(cout), int(pool)))


def extract_conv0(img, chan0, attn, Fh, Fw, use_gamma, patch, w0, scale, shift, cout, relu, y0, canvas=None, canvas_chan=-1):
  """extract_direct + layer 0 of the attention CNN (3x3, BN folded, ReLU, no pool) in one launch (ra_extract_conv0_f32).
  w0 [3,3,4,cout]: the filter in the packed input's channel order; y0 [B,Fh,Fw,cout]."""
  _need_cuda(img, attn, patch, canvas, w0, scale, shift, y0)
  B, H, W, Ci = img.shape
  check(rn.lib().ra_extract_conv0_f32(ptr(img), Ci, chan0, ptr(canvas), int(canvas_chan), ptr(attn), B, H, W, Fh, Fw,
                                      int(use_gamma), ptr(patch), ptr(w0), ptr(scale), ptr(shift), int(cout), int(relu), ptr(y0),
                                      rn.stream_ptr()), 'ra_extract_conv0_f32')


PASTE_Y_PREFILLED, PASTE_CANVAS_FLOORED = 1, 2


def paste_direct(patch, pc, attn, beta, disable_overwrite, y_out, y_stride_b, H, W, canvas=None,
                 img=None, canvas_chan=-1, flags=0):
  _need_cuda(patch, attn, canvas, img)
  B, Fh, Fw, Cp = patch.shape
  Ci = 0 if img is None else img.shape[3]
  check(rn.lib().ra_paste_direct_f32(ptr(patch), Cp, pc, ptr(attn), B, H, W, Fh, Fw, C.c_float(beta),
                                     int(disable_overwrite), ptr(canvas), ptr(img), Ci,
                                     int(canvas_chan), ptr(y_out), y_stride_b, int(flags),
                                     rn.stream_ptr()),
        'ra_paste_direct_f32')


def paste_score_direct(patch, pc, attn, beta, disable_overwrite, y_out, y_stride_b, H, W, canvas, flags,
                       h, core, w, bias, s_out_ptr, s_stride_b):
  """paste_direct on a canvas plane + the score MLP of the timestep in the same launch."""
  _need_cuda(patch, attn, canvas, h, core, w, bias)
  B, Fh, Fw, Cp = patch.shape
  K1 = 0 if core is None else core.shape[1]
  check(rn.lib().ra_paste_score_direct_f32(ptr(patch), Cp, pc, ptr(attn), B, H, W, Fh, Fw, C.c_float(beta),
                                           int(disable_overwrite), ptr(canvas), ptr(y_out), y_stride_b,
                                           int(flags), ptr(h), h.shape[1], ptr(core), K1, ptr(w), ptr(bias),
                                           ptr(s_out_ptr), s_stride_b, rn.stream_ptr()),
        'ra_paste_score_direct_f32')


RESAMPLE_READ, RESAMPLE_WRITE, RESAMPLE_BOX = 0, 1, 2


def resample_bwd(mode, rec, H, W, Fh, Fw, X=None, chan0=0, C=None, dY=None, Y=None, Q=None, E=None, scale=None, div=None):
  """ra_resample_bwd_f32 (include/recattend.h): the adjoint of one attention resample straight to the gradients of the
  window parameters.  Returns out [B,8] = (d ctr_y, d ctr_x, d size_y, d size_x, d lg_var_y, d lg_var_x, d gamma, 0)."""
  _need_cuda(rec, X, dY, Y, Q, E)  # scale / div: strided [B] views (a column of the record), read with their stride
  B = rec.shape[0]
  if mode == RESAMPLE_READ:
    Cx = X.shape[3]
    C = Cx - chan0 if C is None else C
  else:
    Cx, C = 1, 1
  lib = rn.lib()
  nws = lib.ra_resample_bwd_workspace_floats(B, Fh, C)
  ws = torch.empty(nws, dtype=torch.float32, device=rec.device)
  out = torch.empty((B, 8), dtype=torch.float32, device=rec.device)
  sstride = 0 if scale is None else int(scale.stride(0))
  dstride = 0 if div is None else int(div.stride(0))
  check(lib.ra_resample_bwd_f32(int(mode), ptr(X), int(Cx), int(chan0), int(C), ptr(dY), ptr(Y), ptr(rec), ptr(Q),
                                0 if Q is None else int(Q.shape[3]), ptr(E), 0 if E is None else int(E.shape[3]), B, H, W, Fh, Fw,
                                ptr(scale), sstride, ptr(div), dstride, ptr(ws), nws, ptr(out), rn.stream_ptr()),
        'ra_resample_bwd_f32')
  return out


def attn_box_direct(attn, H, W, Fh, Fw, beta, out, stride_b):
  _need_cuda(attn)
  check(rn.lib().ra_attn_box_direct_f32(ptr(attn), attn.shape[0], H, W, Fh, Fw, C.c_float(beta),
                                        ptr(out), stride_b, rn.stream_ptr()),
        'ra_attn_box_direct_f32')


def dense(x0, W, b, act, out, out_stride_b, x1=None):
  """act: None/'relu'/'sigmoid'/'softmax'/'tanh'.  out may be a view; pass its data ptr."""
  _need_cuda(x0, x1, W, b)
  code = {None: 0, 'relu': 1, 'sigmoid': 2, 'softmax': 3, 'tanh': 4}[act]
  K1 = 0 if x1 is None else x1.shape[1]
  check(rn.lib().ra_dense_f32(ptr(x0), x0.shape[1], ptr(x1), K1, ptr(W), ptr(b), x0.shape[0],
                              W.shape[1], code, ptr(out), out_stride_b, rn.stream_ptr()),
        'ra_dense_f32')


def pack_input(x, d_in, y_in, Cp, packed, canvas_plane=None):
  """canvas_plane: the decode loop's separate canvas [B,H,W], zeroed by the same launch."""
  _need_cuda(x, d_in, y_in, packed, canvas_plane)
  B, H, W, D = x.shape
  Dd = 0 if d_in is None else d_in.shape[3]
  Dy = 0 if y_in is None else y_in.shape[3]
  check(rn.lib().ra_pack_input_plane_f32(ptr(x), D, ptr(d_in), Dd, ptr(y_in), Dy, B, H, W, Cp,
                                         ptr(packed), ptr(canvas_plane), rn.stream_ptr()), 'ra_pack_input_plane_f32')


def canvas_max(img, canvas_chan, ysel, noise):
  _need_cuda(img, ysel, noise)
  B, H, W, Ci = img.shape
  check(rn.lib().ra_canvas_max_f32(ptr(img), Ci, canvas_chan, ptr(ysel), ptr(noise), B, H, W,
                                   rn.stream_ptr()), 'ra_canvas_max_f32')


def gaussian_filter(center, size, lg_var, L, F):
  _need_cuda(center, size, lg_var)
  B = center.shape[0]
  out = torch.empty((B, L, F), dtype=torch.float32, device=center.device)
  check(rn.lib().ra_gaussian_filter_f32(ptr(center), ptr(size), ptr(lg_var), B, L, F, ptr(out),
                                        rn.stream_ptr()), 'ra_gaussian_filter_f32')
  return out


def extract_patch_dense(x, f_y, f_x):
  _need_cuda(x, f_y, f_x)
  B, H, W, D = x.shape
  FH, FW = f_y.shape[2], f_x.shape[2]
  out = torch.empty((B, FH, FW, D), dtype=torch.float32, device=x.device)
  check(rn.lib().ra_extract_patch_dense_f32(ptr(x), ptr(f_y), ptr(f_x), B, H, W, D, FH, FW,
                                            ptr(out), rn.stream_ptr()),
        'ra_extract_patch_dense_f32')
  return out


def affine_act(x, scale, shift, relu=False):
  _need_cuda(x, scale, shift)
  Cc = x.shape[-1]
  out = torch.empty_like(x)
  check(rn.lib().ra_affine_act_f32(ptr(x), ptr(scale), ptr(shift), x.numel() // Cc, Cc,
                                   int(relu), ptr(out), rn.stream_ptr()), 'ra_affine_act_f32')
  return out


def max_pool(x, ratio):
  _need_cuda(x)
  B, H, W, Cc = x.shape
  out = torch.empty((B, -(-H // ratio), -(-W // ratio), Cc), dtype=torch.float32,
                    device=x.device)
  check(rn.lib().ra_max_pool_f32(ptr(x), B, H, W, Cc, int(ratio), ptr(out), rn.stream_ptr()),
        'ra_max_pool_f32')
  return out


def hungarian(weights):
  """Drop-in for hungarian_module.hungarian (modellib.py:406): weights [B,N,M] or [N,M] ->
  (matching, cover_x [...,N,1], cover_y [...,1,M]).  CPU tensors/arrays run the host entry
  point (like the reference's CPU op); CUDA tensors run the device kernel."""
  is_t = isinstance(weights, torch.Tensor)
  two_d = (weights.ndim == 2)
  if weights.ndim not in (2, 3):
    raise rn.RecAttendError('Must have dimension 3 or 2.')  # hungarian.cc:62
  if is_t and weights.is_cuda:
    w = weights.detach().to(torch.float32).contiguous()
    w3 = w[None] if two_d else w
    B, N, M = w3.shape
    m = torch.empty_like(w3)
    cx = torch.empty((B, N), dtype=torch.float32, device=w.device)
    cy = torch.empty((B, M), dtype=torch.float32, device=w.device)
    st = torch.zeros((max(B, 1),), dtype=torch.int32, device=w.device)
    nb = rn.lib().ra_hungarian_dev_workspace_bytes(max(B, 1), N, M)
    ws = torch.empty((nb,), dtype=torch.uint8, device=w.device)
    check(rn.lib().ra_hungarian_f32_dev(ptr(w3), B, N, M, ptr(m), ptr(cx), ptr(cy), ptr(st),
                                        ptr(ws), nb, rn.stream_ptr()), 'ra_hungarian_f32_dev')
    hungarian.last_status = st
    check_match_status(st, 'hungarian')
    cx, cy = cx[:, :, None], cy[:, None, :]
    return (m[0], cx[0], cy[0]) if two_d else (m, cx, cy)
  w = _np32(weights)
  w3 = w[None] if two_d else w
  B, N, M = w3.shape
  m = np.zeros_like(w3)
  cx = np.zeros((B, N), np.float32)
  cy = np.zeros((B, M), np.float32)
  rc = rn.lib().ra_hungarian_f32(ptr(w3), B, N, M, ptr(m), ptr(cx), ptr(cy))
  if rc < 0:
    check(rc, 'ra_hungarian_f32')
  hungarian.last_status = rc
  cx, cy = cx[:, :, None], cy[:, None, :]
  if two_d:
    m, cx, cy = m[0], cx[0], cy[0]
  if is_t:
    return torch.from_numpy(m), torch.from_numpy(cx), torch.from_numpy(cy)
  return m, cx, cy


hungarian.last_status = 0


def check_match_status(status, what='Hungarian'):
  """Per-example status of the device solver (int32 [B], include/recattend.h): a negative code is
  where the reference LOG(FATAL)s (hungarian.cc:124-127,146-160,184-188,446-450) — raise instead of
  carrying a partial matching into the loss; 1 is the outer 1000-iteration cap, where the
  reference logs an error and returns the partial matching (hungarian.cc:363-377) — warn."""
  if status is None or isinstance(status, int):
    worst, capped = (status or 0), (status == 1)
  else:
    worst = int(status.min().item())
    capped = bool((status == 1).any().item())
  if worst < 0:
    raise rn.RecAttendError('%s: the matching solver stopped with code %d (an iteration cap the '
                            'reference aborts on, hungarian.cc LOG(FATAL))' % (what, worst))
  if capped:
    import warnings
    warnings.warn('%s: outer iteration cap reached, partial matching returned (hungarian.cc:363-377)'
                  % what)


# --------------------------------------------------------------------------------------
# loss / statistics head (csrc/ra_loss.hip)
# --------------------------------------------------------------------------------------
STAT_NAMES = ('loss', 'box_loss', 'segm_loss', 'conf_loss', 'iou_soft', 'iou_soft_box',
              'wt_cov_soft', 'unwt_cov_soft', 'iou_hard', 'wt_cov_hard', 'unwt_cov_hard', 'dice',
              'count_acc', 'dic', 'dic_abs')  # RA_STAT_* order


def pair_stats(a, b, want=('iou_soft', 'iou_hard', 'dice_hard', 'sum_a', 'sum_b'), a_tmajor=False):
  """One pass over a [B,N,H,W] and b [B,M,H,W]: pairwise soft IoU, IoU / DICE of (a > 0.5), and
  the per-instance sums (modellib.f_iou / f_dice pairwise=True, full_model.py:981,1064-1073).
  a_tmajor: a is stored [N,B,H,W] (the training step's timestep-major masks), read in place through strides."""
  a, b = a.contiguous(), b.contiguous()
  _need_cuda(a, b)
  if a_tmajor:
    N, B, H, W = a.shape
  else:
    B, N, H, W = a.shape
  M = b.shape[1]
  dev = a.device
  n = rn.lib().ra_pair_stats_workspace_floats(B, H * W)
  ws = torch.empty((n,), dtype=torch.float32, device=dev)
  out = {}
  for k, shp in (('iou_soft', (B, N, M)), ('iou_hard', (B, N, M)), ('dice_hard', (B, N, M)),
                 ('sum_a', (B, N)), ('sum_b', (B, M)), ('inter', (B, N, M)), ('sum_a_hard', (B, N))):
    out[k] = torch.empty(shp, dtype=torch.float32, device=dev) if k in want else None
  if a_tmajor:
    check(rn.lib().ra_pair_stats_strided_f32(ptr(a), H * W, B * H * W, ptr(b), B, N, M, H * W, ptr(ws), n, ptr(out['iou_soft']),
                                             ptr(out['iou_hard']), ptr(out['dice_hard']), ptr(out['sum_a']), ptr(out['sum_b']),
                                             ptr(out['inter']), ptr(out['sum_a_hard']), rn.stream_ptr()), 'ra_pair_stats_strided_f32')
    return out
  check(rn.lib().ra_pair_stats_f32(ptr(a), ptr(b), B, N, M, H * W, ptr(ws), n, ptr(out['iou_soft']),
                                   ptr(out['iou_hard']), ptr(out['dice_hard']), ptr(out['sum_a']),
                                   ptr(out['sum_b']), ptr(out['inter']), ptr(out['sum_a_hard']),
                                   rn.stream_ptr()), 'ra_pair_stats_f32')
  return out


def gt_box(y_gt, padding_ratio, min_padding, want_box=True, want_ws=False):
  """modellib.get_gt_box (modellib.py:663-701), center_shift_ratio = 0: params [B,T,8], box (want_ws: also the
  workspace holding the min / max / sum partials, which knob_setup reads)."""
  y_gt = y_gt.contiguous()
  _need_cuda(y_gt)
  B, T, H, W = y_gt.shape
  params = torch.empty((B, T, 8), dtype=torch.float32, device=y_gt.device)
  box = torch.empty((B, T, H, W), dtype=torch.float32, device=y_gt.device) if want_box else None
  n = rn.lib().ra_gt_box_workspace_floats(B, T)
  ws = torch.empty((n,), dtype=torch.float32, device=y_gt.device)
  check(rn.lib().ra_gt_box_f32(ptr(y_gt), B, T, H, W, C.c_float(padding_ratio),
                               C.c_float(min_padding), ptr(ws), n, ptr(params), ptr(box),
                               rn.stream_ptr()), 'ra_gt_box_f32')
  return (params, box, ws) if want_ws else (params, box)


def knob_setup(gt_ws, B, T, pad, shift, u_box, u_segm, sched, min_padding, timescale):
  """The step's noisy ground-truth attention (ctr, size [B,T,2]) and knob masks [B,T,1] from gt_box's partials: one launch
  (full_model.py:567-577,596-625)."""
  pad, shift, u_box, u_segm, sched = (t.contiguous() for t in (pad, shift, u_box, u_segm, sched))
  _need_cuda(gt_ws, pad, shift, u_box, u_segm, sched)
  dev = gt_ws.device
  ctr, size = torch.empty((B, T, 2), dtype=torch.float32, device=dev), torch.empty((B, T, 2), dtype=torch.float32, device=dev)
  kb, ks = torch.empty((B, T, 1), dtype=torch.float32, device=dev), torch.empty((B, T, 1), dtype=torch.float32, device=dev)
  check(rn.lib().ra_knob_setup_f32(ptr(gt_ws), B, T, ptr(pad), ptr(shift), ptr(u_box), ptr(u_segm), ptr(sched), C.c_float(min_padding),
                                   int(bool(timescale)), ptr(ctr), ptr(size), ptr(kb), ptr(ks), rn.stream_ptr()), 'ra_knob_setup_f32')
  return ctr, size, kb, ks


def box_iou_rects(box, params):
  """f_iou (soft) of box [B,H,W] against the T rectangles of gt_box's params [B,T,8] -> [B,T]."""
  box, params = box.contiguous(), params.contiguous()
  _need_cuda(box, params)
  B, H, W = box.shape
  T = params.shape[1]
  out = torch.empty((B, T), dtype=torch.float32, device=box.device)
  ws = torch.empty((rn.lib().ra_box_iou_rects_workspace_floats(B),), dtype=torch.float32, device=box.device)
  check(rn.lib().ra_box_iou_rects_f32(ptr(box), ptr(params), B, T, H, W, ptr(ws), ws.numel(), ptr(out), rn.stream_ptr()),
        'ra_box_iou_rects_f32')
  return out


def segm_match(iou, s_gt):
  """modellib.f_segm_match (modellib.py:382-415) on device; returns (match [B,N,N], status [B])."""
  iou, s_gt = iou.contiguous(), s_gt.contiguous()
  _need_cuda(iou, s_gt)
  B, N, _ = iou.shape
  nb = rn.lib().ra_segm_match_workspace_bytes(B, N)
  ws = torch.empty((nb,), dtype=torch.uint8, device=iou.device)
  match = torch.empty_like(iou)
  status = torch.zeros((B,), dtype=torch.int32, device=iou.device)
  check(rn.lib().ra_segm_match_f32(ptr(iou), ptr(s_gt), B, N, ptr(ws), nb, ptr(match), ptr(status),
                                   rn.stream_ptr()), 'ra_segm_match_f32')
  return match, status


def segm_match_host(iou, s_gt, block=None, threads=16):
  """modellib.f_segm_match with the Hungarian problems solved on HOST cores, side by side, as a host function of the current stream
  (a host node under graph capture): ra_segm_match_host_f32.  block: (pinned uint8 host tensor, device float scratch) kept by the
  caller for as long as a captured graph may replay the call (None: a fresh one is made and returned).  -> (match, status, block)."""
  iou, s_gt = iou.contiguous(), s_gt.contiguous()
  _need_cuda(iou, s_gt)
  B, N, _ = iou.shape
  nb = rn.lib().ra_segm_match_host_block_bytes(B, N)
  if block is None or block[0].numel() < nb or block[1].numel() < B * N * N:
    block = (torch.empty((nb,), dtype=torch.uint8).pin_memory(), torch.empty((B * N * N,), dtype=torch.float32, device=iou.device))
  match = torch.empty_like(iou)
  status = torch.zeros((B,), dtype=torch.int32, device=iou.device)
  check(rn.lib().ra_segm_match_host_f32(ptr(iou), ptr(s_gt), B, N, ptr(block[1]), block[0].data_ptr(), block[0].numel(), int(threads),
                                        ptr(match), ptr(status), rn.stream_ptr()), 'ra_segm_match_host_f32')
  return match, status, block


def loss_stats(iou_soft, iou_hard, dice, match_real, iou_box, match_box, s_out, s_gt, sum_gt,
               fixed_order=False, segm_loss_fn='iou', loss_mix_ratio=1.0):
  """Every scalar of full_model.py:941-1081 -> float32 [len(STAT_NAMES)] on the device."""
  ts = [t.contiguous() for t in (iou_soft, iou_hard, dice, match_real, iou_box, match_box, s_out,
                                 s_gt, sum_gt)]
  _need_cuda(*ts)
  B, T = s_gt.shape
  out = torch.empty((len(STAT_NAMES),), dtype=torch.float32, device=s_gt.device)
  fn = {'iou': 0, 'wt_cov': 1}[segm_loss_fn]
  n = rn.lib().ra_loss_stats_workspace_floats(B)
  ws = torch.empty((n,), dtype=torch.float32, device=s_gt.device)
  check(rn.lib().ra_loss_stats_f32(*[ptr(t) for t in ts], B, T, int(bool(fixed_order)), fn,
                                   C.c_float(loss_mix_ratio), ptr(ws), n, ptr(out), rn.stream_ptr()),
        'ra_loss_stats_f32')
  return out


# --------------------------------------------------------------------------------------
# evaluation post-processing and metrics (csrc/ra_eval.hip)
# --------------------------------------------------------------------------------------
EVAL_NAMES = ('sbd', 'wt_cov', 'unwt_cov', 'fg_iou', 'fg_dice', 'avg_fp', 'avg_fn', 'count_acc',
              'count_mse', 'dic', 'dic_abs', 'num_obj', 'count_out')  # RA_EVAL_* order
EVALI_NAMES = ('obj_pr', 'obj_re', 'pix_pr', 'pix_re', 'has_out', 'is_gt')  # RA_EVALI_* order


def postprocess(y_out, s_out, thresh, fg=None, want_union=False):
  """apply_confidence -> apply_one_label -> apply_threshold [-> mask_foreground] in one pass
  (utils/postprocess.py:5-52,139-147): (y_bin [B,T,H,W], s_hard [B,T], union [B,H,W] | None)."""
  y_out, s_out = y_out.contiguous(), s_out.contiguous()
  fg = None if fg is None else fg.contiguous()
  _need_cuda(y_out, s_out, fg)
  B, T, H, W = y_out.shape
  y_bin = torch.empty_like(y_out)
  s_hard = torch.empty((B, T), dtype=torch.float32, device=y_out.device)
  uni = torch.empty((B, H, W), dtype=torch.float32, device=y_out.device) if want_union else None
  check(rn.lib().ra_postprocess_f32(ptr(y_out), ptr(s_out), B, T, H, W, C.c_float(thresh),
                                    ptr(fg), ptr(y_bin),
                                    ptr(s_hard), ptr(uni), rn.stream_ptr()), 'ra_postprocess_f32')
  return y_bin, s_hard, uni


def union_over_instances(y):
  """y [B,T,H,W] -> max over T [B,H,W] (analysis.py:547,570)."""
  y = y.contiguous()
  _need_cuda(y)
  B, T, H, W = y.shape
  out = torch.empty((B, H, W), dtype=torch.float32, device=y.device)
  check(rn.lib().ra_union_f32(ptr(y), B, T, H * W, ptr(out), rn.stream_ptr()), 'ra_union_f32')
  return out


def dilate(y, radius=2):
  """cv2.dilate(plane, ones((2 radius + 1,) * 2)) on every [H,W] plane of y [..., H, W] (postprocess.py:63-72)."""
  y = y.contiguous()
  _need_cuda(y)
  H, W = y.shape[-2:]
  out = torch.empty_like(y)
  check(rn.lib().ra_dilate_f32(ptr(y), y.numel() // (H * W), H, W, int(radius), ptr(out), rn.stream_ptr()), 'ra_dilate_f32')
  return out


def resize_linear(y, H, W):
  """cv2.resize(plane, (W, H), interpolation=INTER_LINEAR) on every plane of y [..., Hs, Ws] (postprocess.py:103-104)."""
  y = y.contiguous()
  _need_cuda(y)
  Hs, Ws = y.shape[-2:]
  out = torch.empty(y.shape[:-2] + (int(H), int(W)), dtype=torch.float32, device=y.device)
  check(rn.lib().ra_resize_linear_f32(ptr(y), y.numel() // (Hs * Ws), Hs, Ws, int(H), int(W), ptr(out), rn.stream_ptr()),
        'ra_resize_linear_f32')
  return out


def bilateral5(y, sigma_color=10.0, sigma_space=10.0):
  """cv2.bilateralFilter(plane, 5, sigma_color, sigma_space) on every plane of y [..., H, W] (postprocess.py:105)."""
  y = y.contiguous()
  _need_cuda(y)
  H, W = y.shape[-2:]
  out = torch.empty_like(y)
  check(rn.lib().ra_bilateral5_f32(ptr(y), y.numel() // (H * W), H, W, C.c_float(sigma_color), C.c_float(sigma_space), ptr(out),
                                   rn.stream_ptr()), 'ra_bilateral5_f32')
  return out


def remove_tiny(y_bin, sizes, conf, threshold):
  """In place: instance planes of at most `threshold` pixels are zeroed, their conf too."""
  _need_cuda(y_bin, sizes, conf)
  B, T, H, W = y_bin.shape
  check(rn.lib().ra_remove_tiny_f32(ptr(y_bin), ptr(sizes.contiguous()), ptr(conf), B, T, H * W,
                                    C.c_float(threshold), rn.stream_ptr()), 'ra_remove_tiny_f32')


def eval_metrics(y_bin, y_gt, s_gt, fg_a=None, fg_b=None):
  """Every per-image statistic of analysis.py:314-760 for binary outputs y_bin and ground truth
  y_gt [B,T,H,W]: dict(iou_pairwise [B,T,T], stats [B,len(EVAL_NAMES)], inst [B,len(EVALI_NAMES),T])."""
  y_bin, y_gt, s_gt = y_bin.contiguous(), y_gt.contiguous(), s_gt.contiguous()
  _need_cuda(y_bin, y_gt, s_gt)
  B, T, H, W = y_bin.shape
  dev = y_bin.device
  if fg_a is None:
    fg_a = union_over_instances(y_bin)
  if fg_b is None:
    fg_b = union_over_instances(y_gt)
  main = pair_stats(y_bin, y_gt, want=('inter', 'sum_a', 'sum_b'))
  ua, ub = fg_a.view(B, 1, H, W), fg_b.view(B, 1, H, W)
  fg = pair_stats(ua, ub, want=('inter', 'sum_a', 'sum_b'))
  a_fgb = pair_stats(y_bin, ub, want=('inter',))['inter'].view(B, T).contiguous()
  b_fga = pair_stats(ua, y_gt, want=('inter',))['inter'].view(B, T).contiguous()
  iou = torch.empty((B, T, T), dtype=torch.float32, device=dev)
  stats = torch.empty((B, len(EVAL_NAMES)), dtype=torch.float32, device=dev)
  inst = torch.empty((B, len(EVALI_NAMES), T), dtype=torch.float32, device=dev)
  check(rn.lib().ra_eval_metrics_f32(ptr(main['inter']), ptr(main['sum_a']), ptr(main['sum_b']),
                                     ptr(s_gt.contiguous()), ptr(fg['inter']), ptr(fg['sum_a']),
                                     ptr(fg['sum_b']), ptr(a_fgb), ptr(b_fga), B, T, ptr(iou), ptr(stats),
                                     ptr(inst), rn.stream_ptr()), 'ra_eval_metrics_f32')
  return {'iou_pairwise': iou, 'stats': stats, 'inst': inst, 'sizes': main['sum_a']}


def random_transform(x, padding, off_y, off_x, flip_v=False, flip_h=False, transpose=False, out=None):
  """image_ops.random_transformation for given draws on x [N,H,W,C] (or [N,H,W] planes); out: a contiguous tensor of x's
  shape to write into (not x itself)."""
  x = x.contiguous()
  _need_cuda(x, out)
  shape = x.shape
  N, H, W = shape[0], shape[1], shape[2]
  Cc = shape[3] if x.dim() == 4 else 1
  if out is None or tuple(out.shape) != tuple(shape) or not out.is_contiguous() or out.dtype != x.dtype or out.data_ptr() == x.data_ptr():
    out = torch.empty_like(x)
  check(rn.lib().ra_random_transform_f32(ptr(x), N, H, W, Cc, int(padding), int(off_y), int(off_x),
                                         int(bool(flip_v)), int(bool(flip_h)), int(bool(transpose)),
                                         ptr(out), rn.stream_ptr()), 'ra_random_transform_f32')
  return out


def colour_jitter(x, hue_delta, saturation_factor, brightness_delta, contrast_factor):
  """image_ops.py:99-103 for given draws on x [B,H,W,3] (ra_colour_jitter_f32)."""
  x = x.contiguous()
  _need_cuda(x)
  B, H, W, Cc = x.shape
  if Cc != 3:
    raise rn.RecAttendError('colour jitter needs an RGB image (last dimension 3)')
  out = torch.empty_like(x)
  n = rn.lib().ra_colour_jitter_workspace_floats(B)
  ws = torch.empty((n,), dtype=torch.float32, device=x.device)
  check(rn.lib().ra_colour_jitter_f32(ptr(x), B, H * W, C.c_float(hue_delta), C.c_float(saturation_factor), C.c_float(brightness_delta),
                                      C.c_float(contrast_factor), ptr(ws), n, ptr(out), rn.stream_ptr()), 'ra_colour_jitter_f32')
  return out


def fill(t, value):
  """t[...] = value as a library launch (keeps framework kernels out of the captured forward)."""
  _need_cuda(t)
  check(rn.lib().ra_fill_f32(ptr(t), t.numel(), C.c_float(value), rn.stream_ptr()), 'ra_fill_f32')
  return t


def tickets_alloc(slots, device):
  """Scratch for `slots` ticketed launches (ra_tile_tickets_bind, include/recattend.h): zeroed by tickets_bind."""
  n = slots * rn.lib().ra_tile_tickets_slot_bytes() // 4
  return torch.zeros((n,), dtype=torch.float32, device=device)


def tickets_bind(t):
  """Zero the scratch on the current stream and make it current for this thread's persistent conv launches: their
  workgroups draw tiles from per-XCD pools instead of walking a fixed list (robust against company on the GPU).
  Returns True if bound; tickets_unbind() ends it.  The launches must all go to the current stream."""
  _need_cuda(t)
  fill(t, 0.0)
  rc = rn.lib().ra_tile_tickets_bind(ptr(t), t.numel() * 4 // rn.lib().ra_tile_tickets_slot_bytes())
  if rc < 0:
    check(rc, 'ra_tile_tickets_bind')
  return rc == 1


def tickets_unbind():
  rn.lib().ra_tile_tickets_bind(None, 0)


def greedy_match(score, out=None):
  """modellib.f_greedy_match with matched == 0 (modellib.py:365-379): score [B,T] -> [B,T]."""
  score = score.contiguous()
  _need_cuda(score, out)
  if out is None:
    out = torch.empty_like(score)
  check(rn.lib().ra_greedy_match_f32(ptr(score), score.shape[0], score.shape[1], ptr(out),
                                     rn.stream_ptr()), 'ra_greedy_match_f32')
  return out


def weighted_sum(w, y, out):
  """out[b] = sum_t w[b,t] * y[b,t] for y [B,T,H,W] (box_model.py:487-499)."""
  w = w.contiguous()
  _need_cuda(w, y, out)
  B, T, H, W = y.shape
  check(rn.lib().ra_weighted_sum_f32(ptr(w), ptr(y), B, T, H * W, ptr(out), rn.stream_ptr()),
        'ra_weighted_sum_f32')
