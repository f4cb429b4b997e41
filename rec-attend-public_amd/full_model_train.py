#!/usr/bin/env python
"""Entry point with the flag surface of the reference's full_model_train.py (:460-668).

Builds the model from the same flags / `model_opt`, persists `model_opt.yaml` + weights under
results/<model_id>/ (the layout full_model_eval.py restores from, utils/saver.py:7-35 with .npz
instead of a TF checkpoint) and runs the trainer's loop (experiment.py:220-274 ->
Trainer.run_step, full_model_train.py:107): `model.run(['loss', 'train_step'], feed{x, y_gt, s_gt,
phase_train=True})` — forward on BatchNorm batch statistics with the ground-truth knobs, both
Hungarian matchings on the device, backward through the HIP kernels, one RCCL all-reduce of the
flat gradient bucket when launched with torch.distributed.run (one process per GPU, every rank
its own shard of the global --batch_size), clip + Adam, BN EMA (ra_train.TrainStep).

Data: --input (an .npz with x [N,H,W,3], y_gt [N,T,H,W], s_gt [N,T]) or synthetic CVPPP-shaped
batches (8..T-1 ellipses per image, sorted by area); the reference's HDF5 datasets,
plots and CSV loggers are out of scope.  BatchNorm moments are taken over the
rank's shard."""
import argparse
import os
import time

import numpy as np
import yaml

import cmd_args_parser as cap
import full_model
import ra_dist
import ra_train


def build_parser():
  p = argparse.ArgumentParser(description='Train full model (recurrent attention)')
  for table in (cap.TRAIN_FLAGS, cap.DATA_FLAGS, cap.MODEL_FLAGS, cap.LEGACY_FLAGS):
    cap.add_flags(p, table)
  cap.add_size_overrides(p)
  p.add_argument('--init_only', action='store_true',
                 help='write model_opt.yaml + initial weights and stop')
  p.add_argument('--input', default=None, help='.npz with x, y_gt, s_gt (default: synthetic batches)')
  p.add_argument('--seed', type=int, default=1234)
  return p


def synthetic_batch(rng, B, H, W, T):
  """CVPPP-shaped ground truth: K ~ U{min(8,T-1)..T-1} random ellipses per image with
  area 0.5-3 % of the image, sorted by area descending (ins_seg_dataset.py:169-172), s_gt[:K] = 1."""
  x = rng.rand(B, H, W, 3).astype(np.float32)
  y = np.zeros((B, T, H, W), np.float32)
  s = np.zeros((B, T), np.float32)
  yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
  for b in range(B):
    k = rng.randint(min(8, T - 1), T) if T > 1 else 1
    objs = []
    for _ in range(k):
      area = rng.uniform(0.005, 0.03) * H * W
      ratio = rng.uniform(0.5, 2.0)
      ry, rx = np.sqrt(area / np.pi * ratio), np.sqrt(area / np.pi / ratio)
      cy, cx = rng.uniform(ry, H - ry), rng.uniform(rx, W - rx)
      objs.append((((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).astype(np.float32))
    objs.sort(key=lambda m: -m.sum())
    for t, m in enumerate(objs):
      y[b, t], s[b, t] = m, 1.0
      x[b][m > 0] = 0.5 * x[b][m > 0] + 0.5 * rng.rand(3).astype(np.float32)
  return x, y, s


def main(argv=None):
  import torch
  args = build_parser().parse_args(argv)
  model_opt = cap.make_model_opt(args, args.inp_height, args.inp_width, args.timespan)
  rank, world, local_rank = ra_dist.init()
  if torch.cuda.is_available():
    torch.cuda.set_device(local_rank)
  model = full_model.get_model(model_opt, is_training=True)
  model_id = args.model_id or 'full_model'
  folder = os.path.join(args.results, model_id)
  if rank == 0:
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, 'model_opt.yaml'), 'w') as f:
      yaml.safe_dump(model_opt, f)
    np.savez(os.path.join(folder, 'weights.npz'), **model.state_dict_numpy())
    print('wrote %s (model_opt.yaml, weights.npz: %d tensors)' % (folder, len(model.weight_keys())))
  if args.init_only:
    return
  H, W, T = model_opt['inp_height'], model_opt['inp_width'], model_opt['timespan']
  lo, hi = ra_dist.shard_range(rank, world, args.batch_size)
  if hi <= lo:
    raise SystemExit('batch_size %d < world size %d' % (args.batch_size, world))
  data = dict(np.load(args.input)) if args.input else None
  rng = np.random.RandomState(args.seed + 7919 * rank)       # rank-offset streams
  gen = torch.Generator(device='cuda').manual_seed(args.seed + 7919 * rank)
  t0 = time.time()
  for step in range(args.num_steps):
    if data is None:
      x, y_gt, s_gt = synthetic_batch(rng, hi - lo, H, W, T)
    else:
      n = data['x'].shape[0]
      idx = (step * args.batch_size + np.arange(lo, hi)) % n
      x, y_gt, s_gt = data['x'][idx], data['y_gt'][idx], data['s_gt'][idx]
    feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'generator': gen}
    loss, _ = model.run(['loss', 'train_step'], feed)
    if rank == 0 and (step % args.steps_per_log == 0 or step == args.num_steps - 1):
      print('step %d  loss %.5f  learn_rate %.2e  %.2f s' % (step, float(loss), ra_train.learn_rate(model_opt, step),
                                                             time.time() - t0))
    if rank == 0 and args.save_ckpt and (step + 1) % args.steps_per_ckpt == 0:
      np.savez(os.path.join(folder, 'weights.npz'), **model.state_dict_numpy())
  ra_dist.barrier()
  if rank == 0:
    np.savez(os.path.join(folder, 'weights.npz'), **model.state_dict_numpy())
    print('trained %d steps, weights -> %s' % (args.num_steps, folder))


if __name__ == '__main__':
  main()
