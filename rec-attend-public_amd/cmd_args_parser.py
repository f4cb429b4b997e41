"""Flag surface of the reference's CLIs (cmd_args_parser.py:18-206, full_model_train.py:460-658)
as data tables: the same flag names, defaults and types, and the same `model_opt` keys, so the
reference's run scripts (run_cvppp.sh, run_kitti.sh, run_cityscapes.sh) keep their arguments.
Only the model-defining part is acted on; harness flags (logging, plotting, checkpoint cadence,
prefetch threads) are accepted for compatibility (out of scope)."""
import argparse

# dataset -> (inp_height, inp_width, timespan)   (cmd_args_parser.py:18-63)
INP_DIM = {
    'synth_shape': (224, 224, None), 'kitti': (128, 448, 20), 'kitti_flow': (128, 448, 20),
    'cvppp': (224, 224, 21), 'mscoco_person': (224, 224, 23), 'mscoco_zebra': (224, 224, 15),
    'cityscapes': (256, 512, 20),
}

_S, _I, _F, _B = str, int, float, 'flag'
_LIST = 'intlist'

# (flag, kind, default)  — ModelArgsParser.add_args, full_model_train.py:460-550
MODEL_FLAGS = [
    ('padding', _I, 16), ('weight_decay', _F, 5e-5), ('base_learn_rate', _F, 0.001),
    ('learn_rate_decay', _F, 0.96), ('steps_per_learn_rate_decay', _I, 5000),
    ('loss_mix_ratio', _F, 1.0), ('segm_loss_fn', _S, 'iou'), ('mlp_dropout', _F, None),
    ('fixed_order', _B, False), ('add_skip_conn', _B, False),
    ('filter_height', _I, 48), ('filter_width', _I, 48),
    ('ctrl_cnn_filter_size', _LIST, '3,3,3,3,3'), ('ctrl_cnn_depth', _LIST, '4,8,16,16,32'),
    ('ctrl_cnn_pool', _LIST, '2,2,2,2,2'), ('attn_cnn_filter_size', _LIST, '3,3,3'),
    ('attn_cnn_depth', _LIST, '4,8,16'), ('attn_cnn_pool', _LIST, '2,2,2'),
    ('attn_dcnn_filter_size', _LIST, '3,3,3,3'), ('attn_dcnn_depth', _LIST, '16,8,4,1'),
    ('attn_dcnn_pool', _LIST, '2,2,2,1'), ('attn_cnn_skip', _S, '1,1,1'),
    ('ctrl_rnn_hid_dim', _I, 256), ('num_ctrl_mlp_layers', _I, 1), ('ctrl_mlp_dim', _I, 256),
    ('box_loss_fn', _S, 'iou'), ('attn_box_padding_ratio', _F, 0.2), ('use_knob', _B, False),
    ('knob_decay', _F, 0.9), ('steps_per_knob_decay', _I, 300), ('knob_base', _F, 1.0),
    ('knob_box_offset', _I, 300), ('knob_segm_offset', _I, 500), ('knob_use_timescale', _B, False),
    ('gt_box_ctr_noise', _F, 0.05), ('gt_box_pad_noise', _F, 0.1), ('gt_segm_noise', _F, 0.3),
    ('clip_gradient', _F, 1.0), ('squash_ctrl_params', _B, False), ('fixed_gamma', _B, False),
    ('pretrain_ctrl_net', _S, None), ('pretrain_attn_net', _S, None), ('pretrain_net', _S, None),
    ('freeze_ctrl_cnn', _B, False), ('freeze_ctrl_rnn', _B, False), ('freeze_ctrl_mlp', _B, False),
    ('freeze_attn_net', _B, False), ('num_ctrl_rnn_iter', _I, 5), ('num_glimpse_mlp_layers', _I, 2),
    ('stop_canvas_grad', _B, False), ('fixed_var', _B, False), ('dynamic_var', _B, False),
    ('use_iou_box', _B, False), ('disable_overwrite', _B, False), ('add_d_out', _B, False),
    ('add_y_out', _B, False), ('num_semantic_classes', _I, 1),
    ('ctrl_add_inp', _B, False), ('ctrl_add_canvas', _B, False), ('ctrl_add_d_out', _B, False),
    ('ctrl_add_y_out', _B, False), ('attn_add_inp', _B, False), ('attn_add_canvas', _B, False),
    ('attn_add_d_out', _B, False), ('attn_add_y_out', _B, False), ('finetune', _B, False),
]
# parsed for compatibility but never read by make_opt (the legacy ConvLSTM-era flags,
# full_model_train.py:463-478)
LEGACY_FLAGS = [
    ('cnn_filter_size', _S, '3,3,3,3,3'), ('cnn_depth', _S, '4,8,8,12,16'), ('cnn_pool', _S, '2,2,2,2,2'),
    ('dcnn_filter_size', _S, '3,3,3,3,3,3'), ('dcnn_depth', _S, '8,6,4,4,2,1'),
    ('dcnn_pool', _S, '2,2,2,2,2,1'), ('rnn_type', _S, 'lstm'), ('conv_lstm_filter_size', _I, 3),
    ('conv_lstm_hid_depth', _I, 12), ('rnn_hid_dim', _I, 256), ('score_maxpool', _I, 1),
    ('num_mlp_layers', _I, 2), ('mlp_depth', _I, 6), ('use_deconv', _B, False),
    ('score_use_core', _B, False),
]
TRAIN_FLAGS = [  # TrainArgsParser, cmd_args_parser.py:93-114
    ('model_id', _S, None), ('num_steps', _I, 500000), ('steps_per_ckpt', _I, 1000),
    ('steps_per_valid', _I, 50), ('steps_per_trainval', _I, 50), ('steps_per_plot', _I, 500),
    ('steps_per_log', _I, 10), ('batch_size', _I, 32), ('results', _S, 'results'),
    ('logs', _S, 'logs'), ('localhost', _S, 'localhost'), ('restore', _S, None),
    ('num_samples_plot', _I, 5), ('save_ckpt', _B, False), ('no_valid', _B, False),
    ('num_batch_valid', _I, 10), ('h5_fname_train', _S, None), ('h5_fname_valid', _S, None),
    ('prefetch', _B, False), ('queue_size', _I, 50), ('num_worker', _I, 4),
]
EVAL_FLAGS = [  # EvalArgsParser + MyEvalArgsParser, cmd_args_parser.py:143-151, full_model_eval.py:180-186
    ('model_id', _S, None), ('batch_size', _I, 32), ('results', _S, './results'), ('output', _S, None),
    ('split', _S, 'valid'), ('prefetch', _B, False), ('queue_size', _I, 50), ('num_worker', _I, 4),
    ('foreground_folder', _S, None), ('threshold_list', _S, None), ('analyzers', _S, None),
    ('test', _B, False), ('no_morph', _B, False), ('remove_tiny', _I, 0),
]
DATA_FLAGS = [('dataset', _S, 'cvppp'), ('dataset_folder', _S, None)]  # DataArgsParser :168-171


def add_flags(parser, table):
  for name, kind, default in table:
    if kind == _B:
      parser.add_argument('--' + name, action='store_true')
    elif kind == _LIST:
      parser.add_argument('--' + name, default=default)
    else:
      parser.add_argument('--' + name, default=default, type=kind)


def make_model_opt(args, inp_height=None, inp_width=None, timespan=None):
  """args -> the model_opt dict of full_model_train.py:581-658 (same keys, same quirks:
  `attn_cnn_skip` ends up as the raw flag string, the rnd_* flips are forced off)."""
  h, w, t = INP_DIM[args.dataset]
  opt = {'inp_height': inp_height or h, 'inp_width': inp_width or w, 'inp_depth': 3,
         'timespan': timespan or t, 'use_bn': True, 'rnd_hflip': False, 'rnd_vflip': False,
         'rnd_transpose': False, 'rnd_colour': False}
  for name, kind, _ in MODEL_FLAGS:
    v = getattr(args, name)
    if kind == _LIST:
      v = [int(s) for s in v.split(',')]
    opt[name] = v
  return opt


def add_size_overrides(parser):
  """Extensions (not in the reference): run the same graph at another resolution / length,
  e.g. BASELINE.json's 512x512, T=16 scale-up of the CVPPP shapes."""
  parser.add_argument('--inp_height', type=int, default=None)
  parser.add_argument('--inp_width', type=int, default=None)
  parser.add_argument('--timespan', type=int, default=None)
