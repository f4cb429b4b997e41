"""ctypes binding of librecattend.so (the C ABI declared in include/recattend.h).

There is NO fallback: if the shared library is missing or a symbol is absent the import
of the product path fails loudly.  Device pointers are taken from torch tensors
(`tensor.data_ptr()`), the stream from `torch.cuda.current_stream()`; torch is plumbing
(memory + streams) only.
"""
import ctypes as C
import os

# The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues
# (default 4, the null stream and the graph-capture stream included) and reads the variable at its
# first HIP call.  full_model.DecodePipeline keeps 4 batches in flight on 4 streams; two of them on one
# queue serialise (3.9 instead of 3.1 ms per cfg2 batch), so ask for 8 unless the user chose.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'librecattend.so')

RA_ABI_VERSION = 113  # include/recattend.h: RA_ABI_VERSION
RA_CONV_TRANSPOSED = 1
RA_E_INVALID, RA_E_SHAPE, RA_E_WORKSPACE = -1, -2, -3  # include/recattend.h
RA_ATTN_STRIDE = 16


class RecAttendError(RuntimeError):
  pass


class CtrlDesc(C.Structure):
  """struct ra_ctrl_desc (include/recattend.h)."""
  _fields_ = [(n, C.c_int) for n in ('G', 'Cf', 'hid', 'iters', 'n_gmlp', 'n_cmlp', 'mlp_dim',
                                      'H', 'W', 'Fh', 'Fw', 'squash', 'fixed_var',
                                      'dynamic_var', 'fixed_gamma')]


_P = C.c_void_p
_I = C.c_int
_F = C.c_float
_Z = C.c_size_t

# name -> (restype, argtypes); must list every symbol of include/recattend.h
SIGNATURES = {
    'ra_version': (_I, []),
    'ra_last_error_string': (C.c_char_p, []),
    'ra_debug_poison_lds': (_I, [_P]),
    'ra_debug_park_xcd': (_I, [_I, _I, _I, _I, _P, _P]),
    'ra_gather_f32': (_I, [_P, _P, _Z, _P, _P]),
    'ra_gemm_tn_acc_f32': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _I, _P]),
    'ra_conv3x3_bf16_f32': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P, _P, _Z, _P, _I, _P]),
    'ra_bn_act_pool_bf16_f32': (_I, [_P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    'ra_bn_act_pool_bwd_acc_bf16_f32': (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P, _P, _P, _P, _I, _P]),
    'ra_bn_act_pool_bwd_grouped_bf16_f32': (_I, [_P, _P, _P, _I, _F, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P, _P, _I, _P]),
    'ra_conv3x3_wgrad_acc_bf16_f32': (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _Z, _P, _I, _I, _P, _P, _I, _P]),
    'ra_colour_jitter_workspace_floats': (_Z, [_I]),
    'ra_colour_jitter_f32': (_I, [_P, _I, _I, _F, _F, _F, _F, _P, _Z, _P, _P]),
    'ra_hungarian_f32': (_I, [_P, _I, _I, _I, _P, _P, _P]),
    'ra_hungarian_dev_workspace_bytes': (_Z, [_I, _I, _I]),
    'ra_hungarian_f32_dev': (_I, [_P, _I, _I, _I, _P, _P, _P, _P, _P, _Z, _P]),
    'ra_conv_cout_padded': (_I, [_I]),
    'ra_conv_packed_floats': (_Z, [_I, _I]),
    'ra_conv_pack_weights': (_I, [_P, _I, _I, _I, _P, _I, _P]),
    'ra_conv_fold_bn': (_I, [_P, _P, _P, _P, _P, _I, _F, _P, _P]),
    'ra_conv3x3_f32': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P]),
    'ra_conv3x3_moments_part_floats': (_Z, [_I]),
    'ra_conv3x3_moments_f32': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P, _P, _Z, _P, _P]),
    'ra_bn_moments_from_partials_f32': (_I, [_P, _I, _I, _P, _P, _P]),
    'ra_conv3x3_bf16ops_f32': (_I, [_P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P]),
    'ra_conv_pair_supported': (_I, [_I, _I, _I]),
    'ra_conv_pair_f32': (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P]),
    'ra_ctrl_packed_floats': (_Z, [C.POINTER(CtrlDesc)]),
    'ra_ctrl_pack_weights': (_I, [C.POINTER(CtrlDesc), _P, _P, _P, _P]),
    'ra_controller_f32': (_I, [C.POINTER(CtrlDesc), _P, _P, _I, _P, _P, _P, _P, _P]),
    'ra_ctrl_split_supported': (_I, [C.POINTER(CtrlDesc)]),
    'ra_ctrl_split_packed_floats': (_Z, [C.POINTER(CtrlDesc)]),
    'ra_ctrl_split_workspace_bytes': (_Z, [C.POINTER(CtrlDesc), _I]),
    'ra_ctrl_split_pack_weights': (_I, [C.POINTER(CtrlDesc), _P, _P, _P, _P]),
    'ra_controller_split_f32': (_I, [C.POINTER(CtrlDesc), _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P, _P]),
    'ra_ctrl_batch_group_images': (_I, [C.POINTER(CtrlDesc), _I]),
    'ra_ctrl_batch_supported': (_I, [C.POINTER(CtrlDesc)]),
    'ra_ctrl_batch_workspace_bytes': (_Z, [C.POINTER(CtrlDesc), _I]),
    'ra_controller_batch_f32': (_I, [C.POINTER(CtrlDesc), _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P, _P]),
    'ra_controller_batch_xcd_f32': (_I, [C.POINTER(CtrlDesc), _P, _P, _I, _P, _P, _P, _P, _P, _Z, _P, _I, _P]),
    'ra_gaussian_filter_f32': (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    'ra_extract_direct_f32': (_I, [_P, _I, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'ra_extract_conv0_supported': (_I, [_I, _I, _I, _I, _I]),
    'ra_extract_conv0_f32': (_I, [_P, _I, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I, _I, _P, _P]),
    'ra_paste_direct_f32': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _I, _I, _P, _Z, _I, _P]),
    'ra_attn_box_direct_f32': (_I, [_P, _I, _I, _I, _I, _I, _F, _P, _Z, _P]),
    'ra_ctrl_train_supported': (_I, [_I, _I, _I, _I, _I]),
    'ra_ctrl_train_save_floats': (_Z, [_I, _I, _I, _I]),
    'ra_ctrl_train_supported_n': (_I, [_I] * 8),
    'ra_ctrl_train_save_floats_n': (_Z, [_I] * 5),
    'ra_ctrl_train_fwd_n_f32': (_I, [_I] * 9 + [_P] * 12),
    'ra_ctrl_train_bwd_n_f32': (_I, [_I] * 9 + [_P] * 12 + [_Z, _P, _Z, _P]),
    'ra_ctrl_train_fwd_f32': (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'ra_ctrl_train_bwd_f32': (_I, [_I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'ra_resample_bwd_workspace_floats': (_Z, [_I, _I, _I]),
    'ra_resample_bwd_f32': (_I, [_I, _P, _I, _I, _I, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _Z, _P, _P]),
    'ra_extract_patch_dense_f32': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    'ra_dense_f32': (_I, [_P, _I, _P, _I, _P, _P, _I, _I, _I, _P, _Z, _P]),
    'ra_pack_input_f32': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P]),
    'ra_pack_input_plane_f32': (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    'ra_canvas_max_f32': (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _P]),
    'ra_affine_act_f32': (_I, [_P, _P, _P, _Z, _I, _I, _P, _P]),
    'ra_max_pool_f32': (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    'ra_pair_stats_workspace_floats': (_Z, [_I, _I]),
    'ra_pair_stats_f32': (_I, [_P, _P, _I, _I, _I, _I, _P, _Z, _P, _P, _P, _P, _P, _P, _P, _P]),
    'ra_pair_stats_strided_f32': (_I, [_P, _Z, _Z, _P, _I, _I, _I, _I, _P, _Z, _P, _P, _P, _P, _P, _P, _P, _P]),
    'ra_gt_box_workspace_floats': (_Z, [_I, _I]),
    'ra_gt_box_f32': (_I, [_P, _I, _I, _I, _I, _F, _F, _P, _Z, _P, _P, _P]),
    'ra_knob_setup_f32': (_I, [_P, _I, _I, _P, _P, _P, _P, _P, _F, _I, _P, _P, _P, _P, _P]),
    'ra_segm_match_workspace_bytes': (_Z, [_I, _I]),
    'ra_segm_match_f32': (_I, [_P, _P, _I, _I, _P, _Z, _P, _P, _P]),
    'ra_segm_match_host_block_bytes': (_Z, [_I, _I]),
    'ra_segm_match_host_f32': (_I, [_P, _P, _I, _I, _P, _P, _Z, _I, _P, _P, _P]),
    'ra_loss_stats_workspace_floats': (_Z, [_I]),
    'ra_loss_stats_f32': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _Z, _P, _P]),
    'ra_postprocess_f32': (_I, [_P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P, _P]),
    'ra_union_f32': (_I, [_P, _I, _I, _I, _P, _P]),
    'ra_dilate_f32': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'ra_resize_linear_f32': (_I, [_P, _I, _I, _I, _I, _I, _P, _P]),
    'ra_bilateral5_f32': (_I, [_P, _I, _I, _I, _F, _F, _P, _P]),
    'ra_remove_tiny_f32': (_I, [_P, _P, _P, _I, _I, _I, _F, _P]),
    'ra_eval_metrics_f32': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    'ra_random_transform_f32': (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'ra_weighted_sum_f32': (_I, [_P, _P, _I, _I, _I, _P, _P]),
    'ra_fill_f32': (_I, [_P, _Z, _F, _P]),
    'ra_tile_tickets_bind': (_I, [_P, _I]),
    'ra_tile_tickets_slot_bytes': (_I, []),
    'ra_adam_step_f32': (_I, [_P, _P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _F, _P]),
    'ra_adam_step_guarded_f32': (_I, [_P, _P, _P, _P, _P, _Z, _F, _F, _F, _F, _F, _F, _P, _I, _I, _P]),
    'ra_bn_workspace_floats': (_Z, [_I]),
    'ra_bn_moments_f32': (_I, [_P, _Z, _I, _P, _Z, _P, _P, _P]),
    'ra_bn_act_pool_f32': (_I, [_P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _P, _P]),
    'ra_bn_act_pool_bwd_f32': (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P, _P, _P]),
    'ra_bn_act_pool_bwd_reduce_f32': (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P, _P, _P, _P]),
    'ra_bn_act_pool_bwd_dx_f32': (_I, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_double, _F, _I, _I, _I, _I, _I, _I, _P, _P]),
    'ra_conv_pack_weights_dev': (_I, [_P, _I, _I, _I, _P, _I, _P, _P]),
    'ra_conv3x3_wgrad_workspace_floats': (_Z, [_I, _I, _I, _I, _I]),
    'ra_conv3x3_wgrad_f32': (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _Z, _P, _P, _P]),
    'ra_conv3x3_wgrad_bf16ops_f32': (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _Z, _P, _P, _P]),
    'ra_conv_wino_supported': (_I, [_I, _I, _I, _I, _I]),
    'ra_conv_wino_packed_floats': (_Z, [_I, _I]),
    'ra_conv_wino_pack_weights': (_I, [_P, _I, _I, _P]),
    'ra_conv_wino_f32': (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P, _P]),
    'ra_conv_split_supported': (_I, [_I, _I, _I, _I, _I]),
    'ra_conv_split_packed_halfs': (_Z, [_I, _I]),
    'ra_conv_split_pack_weights': (_I, [_P, _I, _I, _P]),
    'ra_conv_split_pack_weights_dev': (_I, [_P, _I, _I, _I, _P, _P]),
    'ra_conv_split_f32': (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _I, _I, _P, _P]),
    'ra_conv_split_plane_f32': (_I, [_P, _I, _I, _I, _I, _P, _I, _P, _P, _P, _I, _I, _I, _P, _P]),
    'ra_conv_pair_wino_supported': (_I, [_I, _I, _I, _I, _I, _I]),
    'ra_conv_pair_wino_f32': (_I, [_P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P]),
    'ra_gauss_filter_f32': (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    'ra_gauss_filter_bwd_f32': (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P]),
    'ra_loss_head_f32': (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _P, _P]),
    'ra_loss_head_bwd_f32': (_I, [_P] * 10 + [_I, _I, _I, _F] + [_P] * 6),
    'ra_attn_head_f32': (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'ra_attn_head_rec_f32': (_I, [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    'ra_attn_head_bwd_f32': (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    'ra_knob_mix_f32': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P]),
    'ra_knob_mix_rec_f32': (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P]),
    'ra_knob_mix_bwd_f32': (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    'ra_gauss_filter_strided_f32': (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P]),
    'ra_gauss_filter_strided_bwd_f32': (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P]),
    'ra_lstm_cell_f32': (_I, [_P, _P, _I, _I, _P, _P, _P, _P]),
    'ra_lstm_cell_bwd_f32': (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P]),
    'ra_conv3x3_wgrad_acc_f32': (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _Z, _P, _I, _I, _P, _P, _P]),
    'ra_box_iou_rects_workspace_floats': (_Z, [_I]),
    'ra_box_iou_rects_f32': (_I, [_P, _P, _I, _I, _I, _I, _P, _Z, _P, _P]),
    'ra_bn_act_pool_bwd_grouped_f32': (_I, [_P, _P, _P, _I, _F, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P, _P, _P]),
    'ra_ptr_table': (_I, [_P, _I, _P, _P]),
    'ra_conv3x3_wgrad_multi_acc_f32': (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _I, _I, _P, _P, _I, _P]),
    'ra_conv3x3_wgrad_acc_bf16ops_f32': (_I, [_P, _I, _I, _I, _I, _I, _P, _I, _P, _Z, _P, _I, _I, _P, _P, _P]),
    'ra_bn_act_pool_bwd_acc_f32': (_I, [_P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _P, _P, _P, _P, _P]),
    'ra_canvas_step_f32': (_I, [_P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _I, _P, _P]),
    'ra_subsample_odd_f32': (_I, [_P, _I, _I, _I, _I, _P, _P]),
    'ra_weighted_sum_multi_f32': (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _P]),
    'ra_weighted_sum_multi_strided_f32': (_I, [_P, _P, _P, _I, _I, _I, _I, _P, _Z, _Z, _P]),
    'ra_conv_first_cache_supported': (_I, [_I, _I, _I, _I, _I, _I]),
    'ra_conv_first_cache_floats': (_Z, [_I, _I, _I]),
    'ra_conv_first_cache_f32': (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _P]),
    'ra_conv_pair_fill_cache_f32': (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P]),
    'ra_conv_pair_fill_cache_rider_f32': (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P, _P, _Z, _F, _P]),
    'ra_conv_pair_cached_f32': (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _I, _I, _P, _P]),
    'ra_greedy_match_f32': (_I, [_P, _I, _I, _P, _P]),
    'ra_paste_score_direct_f32': (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P, _Z, _I, _P, _I, _P, _I, _P, _P, _P, _Z, _P]),
}

_lib = None


def lib():
  """The loaded library; raises RecAttendError if it (or any symbol) is missing."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise RecAttendError(
          'librecattend.so not found at %s — build it with `python -c "import '
          '__graft_entry__ as g; g.build()"` (no CPU fallback exists)' % LIB_PATH)
    handle = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
      try:
        fn = getattr(handle, name)
      except AttributeError:
        raise RecAttendError('librecattend.so lacks symbol %s' % name)
      fn.restype = res
      fn.argtypes = args
    if handle.ra_version() != RA_ABI_VERSION:
      raise RecAttendError('librecattend.so has ABI version %d, this binding was written against %d: rebuild it '
                           '(__graft_entry__.build())' % (handle.ra_version(), RA_ABI_VERSION))
    _lib = handle
  return _lib


def check(rc, what):
  """Turn a non-zero return code into an exception carrying ra_last_error_string()."""
  if rc != 0:
    msg = lib().ra_last_error_string()
    raise RecAttendError('%s failed with code %d: %s' % (what, rc, (msg or b'').decode()))


def ptr(t):
  """Raw data pointer of a torch tensor / numpy array / None."""
  if t is None:
    return None
  if isinstance(t, int):
    return t
  if hasattr(t, 'data_ptr'):
    return t.data_ptr()
  return t.ctypes.data


def stream_ptr():
  import torch
  return torch.cuda.current_stream().cuda_stream


class quiet_capture(object):
  """Context for a HIP-graph capture: Python's cyclic garbage collector is parked for its duration.  A
  collection that happens to run inside the capture can destroy an old graph, event or pinned buffer,
  and the runtime aborts the process on such a call while a stream is capturing (seen once the test
  suite ran decode pipelines before a training-step capture)."""

  def __enter__(self):
    import gc
    gc.collect()
    self._was = gc.isenabled()
    gc.disable()
    return self

  def __exit__(self, *exc):
    import gc
    if self._was:
      gc.enable()
    return False
