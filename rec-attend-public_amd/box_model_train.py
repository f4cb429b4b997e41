#!/usr/bin/env python
"""Entry point with the flag surface of the reference's box_model_train.py (:340-462): trains the
controller-only box model (box_model.py:11-669) — the first stage of the reference's two-stage
recipe (run_cvppp.sh:16-28; run_kitti.sh, run_cityscapes.sh) — and writes
results/<model_id>/{model_opt.yaml, weights.npz}.  The archive is what `box_model_read.py` exported
as weights.h5 (:31-60): pass it to `full_model_train.py --pretrain_ctrl_net <weights.npz>` to start
the full model from the pre-trained controller (full_model.py:271-284,315-329,355-366,387-398).

Each step is `model.run(['loss', 'train_step'], feed{x, y_gt, s_gt, phase_train=True})`
(box_model_train.py:114): forward on BatchNorm batch statistics with the teacher-forced noisy canvas,
Hungarian matching of the attention boxes on the device, backward through the HIP kernels, one
all-reduce of the flat gradient bucket across the ranks, clip + Adam (ra_train.BoxTrainStep)."""
import argparse
import os
import time

import numpy as np
import yaml

import box_model
import cmd_args_parser as cap
import full_model_train as fmt
import ra_dist
import ra_train

# the model_opt keys box_model_train.make_opt builds (box_model_train.py:397-451)
BOX_KEYS = ('timespan', 'inp_height', 'inp_width', 'inp_depth', 'padding', 'filter_height', 'filter_width',
            'ctrl_cnn_filter_size', 'ctrl_cnn_depth', 'ctrl_cnn_pool', 'ctrl_rnn_hid_dim', 'num_ctrl_mlp_layers',
            'ctrl_mlp_dim', 'attn_box_padding_ratio', 'weight_decay', 'use_bn', 'box_loss_fn', 'base_learn_rate',
            'learn_rate_decay', 'steps_per_learn_rate_decay', 'pretrain_net', 'squash_ctrl_params', 'clip_gradient',
            'fixed_order', 'num_ctrl_rnn_iter', 'num_glimpse_mlp_layers', 'fixed_var', 'use_iou_box', 'dynamic_var',
            'add_d_out', 'add_y_out', 'rnd_hflip', 'rnd_vflip', 'rnd_transpose', 'rnd_colour', 'num_semantic_classes')


def build_parser():
  p = argparse.ArgumentParser(description='Train box model (controller pre-training)')
  for table in (cap.TRAIN_FLAGS, cap.DATA_FLAGS, cap.MODEL_FLAGS):
    cap.add_flags(p, table)
  cap.add_size_overrides(p)
  p.add_argument('--input', default=None, help='.npz with x, y_gt, s_gt (default: synthetic batches)')
  p.add_argument('--seed', type=int, default=1234)
  p.add_argument('--sync_bn', action='store_true', help='BatchNorm batch moments over the whole data-parallel batch')
  return p


def main(argv=None):
  import torch
  args = build_parser().parse_args(argv)
  full = cap.make_model_opt(args, args.inp_height, args.inp_width, args.timespan)
  model_opt = {k: full[k] for k in BOX_KEYS if k in full}
  model_opt['attn_box_padding_ratio'], model_opt['weight_decay'], model_opt['use_bn'] = 0.2, 5e-5, True  # :421-423
  # 'fixed_var' is args.fixed_var as the reference writes it (box_model_train.py:442; False unless the flag is given), so
  # that the controller is pre-trained with the variance parameterisation full_model then continues with
  model_opt['sync_bn'], model_opt['seed'] = bool(args.sync_bn), int(args.seed)
  rank, world, local_rank = ra_dist.init()
  if torch.cuda.is_available():
    torch.cuda.set_device(local_rank)
  model = box_model.get_model(model_opt)
  folder = os.path.join(args.results, args.model_id or 'box_model')
  H, W, T = model_opt['inp_height'], model_opt['inp_width'], model_opt['timespan']
  lo, hi = ra_dist.shard_range(rank, world, args.batch_size)
  if hi <= lo:
    raise SystemExit('batch_size %d < world size %d' % (args.batch_size, world))
  data = dict(np.load(args.input)) if args.input else None

  add_d = bool(model_opt.get('add_d_out', False))  # run_kitti.sh:45-59: stage 1 trains with --add_d_out --add_y_out
  nsc = int(model_opt.get('num_semantic_classes', 1))

  def make_batch(step):
    if data is None:
      rng = np.random.RandomState(fmt.step_seed(args.seed, rank, step))
      b = fmt.synthetic_batch(rng, hi - lo, H, W, T)
      return b + (dict(zip(('d_in', 'y_in'), fmt.synthetic_extras(rng, hi - lo, H, W, nsc))),) if add_d else b
    idx = (step * args.batch_size + np.arange(lo, hi)) % data['x'].shape[0]
    b = (data['x'][idx], data['y_gt'][idx], data['s_gt'][idx])
    return b + ({'d_in': data['d_in'][idx], 'y_in': data['y_in'][idx]},) if add_d else b

  fmt.train_loop(args, model, model_opt, folder, rank, world, make_batch)


if __name__ == '__main__':
  main()
