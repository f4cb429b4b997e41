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
rank's shard, or over the whole batch with --sync_bn.  Checkpoints (weights.npz +
optim.npz: Adam slots and global_step, utils/saver.py:24-31) are written every --steps_per_ckpt steps and
at the end; --restore <folder> continues from one (experiment.py:26-37)."""
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
  p.add_argument('--save_rank_weights', action='store_true',
                 help='every rank also writes weights_rank<r>.npz (its own weights, EMA shadows and per-step losses) at the end: '
                      'the data-parallel ranks must hold ONE model — tests/test_distributed_gpu.py checks they do')
  p.add_argument('--sync_bn', action='store_true',
                 help='data parallel: BatchNorm batch moments over the WHOLE batch (nnlib.py:98), one small collective per BN call')
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


def synthetic_extras(rng, B, H, W, num_semantic_classes):
  """d_in / y_in of the KITTI / Cityscapes architectures as SURVEY.md §8d defines the synthetic inputs (the reference
  feeds fg_model's outputs, data_api/*): d_in = one-hot over the 8 orientation classes of a seeded integer map, y_in =
  softmax of seeded logits over the semantic classes."""
  d_in = np.eye(8, dtype=np.float32)[rng.randint(0, 8, (B, H, W))]
  z = rng.randn(B, H, W, num_semantic_classes).astype(np.float32)
  e = np.exp(z - z.max(axis=3, keepdims=True))
  return d_in, (e / e.sum(axis=3, keepdims=True)).astype(np.float32)


def save_checkpoint(path, model):
  """Everything utils/saver.py:24-31 saves (tf.all_variables()): the weights with their BN EMA shadows, and — once a
  trainer exists — the Adam slots and global_step.  One .npz; `path` without the optimizer part stays loadable by
  Model.load_weights (the keys are prefixed)."""
  out = dict(model.state_dict_numpy())
  tr = getattr(model, 'trainer', None)
  if tr is not None:
    out.update({'optim/' + k: v for k, v in tr.state_dict().items()})
  else:
    out['optim/global_step'] = np.asarray(int(model.get('global_step', 0) or 0), dtype=np.int64)
  tmp = path + '.tmp.npz'  # written beside and renamed: a crash mid-write must not cost the only checkpoint
  np.savez(tmp, **out)
  os.replace(tmp, path)


def load_checkpoint(path, model):
  """Restore what save_checkpoint wrote into `model` (building its trainer): weights, EMA, Adam m / v, global_step."""
  import ra_train
  data = dict(np.load(path))
  optim = {k[len('optim/'):]: v for k, v in data.items() if k.startswith('optim/')}
  model.load_weights({k: v for k, v in data.items() if not k.startswith('optim/')})
  if getattr(model, 'trainer', None) is None:
    model.trainer = (ra_train.BoxTrainStep if model.box_model else ra_train.TrainStep)(model)
  else:  # the bucket owns the weights' storage: load_weights above wrote through the views
    model.trainer._graphs = {}
  model.trainer.load_state_dict(optim, strict=len(optim) > 1)
  model.trainer.broadcast_state()
  return model


def step_seed(seed, rank, step, stream=0):
  """One seed per (run seed, rank, global step, stream): the random streams are functions of the step."""
  return (int(seed) + 7919 * int(rank) + 104729 * int(step) + 15485863 * int(stream)) % (2 ** 31 - 1)


def train_loop(args, model, model_opt, folder, rank, world, make_batch):
  """experiment.py:220-274 -> Trainer.run_step (full_model_train.py:107), shared with box_model_train.py."""
  import torch
  ckpt = os.path.join(folder, 'weights.npz')
  if getattr(args, 'restore', None):
    src = args.restore if args.restore.endswith('.npz') else os.path.join(args.restore, 'weights.npz')
    load_checkpoint(src, model)
    if rank == 0:
      os.makedirs(folder, exist_ok=True)  # the periodic checkpoints go to `folder`, which need not be where --restore points
      if not os.path.exists(os.path.join(folder, 'model_opt.yaml')):
        with open(os.path.join(folder, 'model_opt.yaml'), 'w') as f:
          yaml.safe_dump(model_opt, f)
      print('restored %s at global_step %d' % (src, model.trainer.bucket.global_step))
  elif rank == 0:
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, 'model_opt.yaml'), 'w') as f:
      yaml.safe_dump(model_opt, f)
    save_checkpoint(ckpt, model)
    print('wrote %s (model_opt.yaml, weights.npz: %d tensors)' % (folder, len(model.weight_keys())))
  if getattr(args, 'init_only', False):
    return
  if args.batch_size % world:
    raise SystemExit('batch_size %d is not a multiple of the world size %d (equal shards: the gradient is averaged '
                     'as sum / world)' % (args.batch_size, world))
  gen = torch.Generator(device='cuda')
  losses = []
  if getattr(model, 'trainer', None) is None:  # before the loop, as load_checkpoint does: the first step's augmentation stream
    model.trainer = (ra_train.BoxTrainStep if model.box_model else ra_train.TrainStep)(model)  # is seeded like every other's
  start = int(model.get('global_step', 0) or 0)
  t0 = time.time()
  for step in range(start, args.num_steps):
    # every random stream of a step — knob draws, crop offset / flips (and make_batch's synthetic data) — is seeded from
    # (seed, rank, step), rank-offset as SURVEY.md §8e asks: a run restored at step k continues the streams of the
    # uninterrupted run instead of replaying those of its first steps
    gen.manual_seed(step_seed(args.seed, rank, step, 1))
    tr = getattr(model, 'trainer', None)
    if tr is not None and getattr(tr, 'aug_gen', None) is not None:
      tr.aug_gen.manual_seed(step_seed(args.seed, rank, step, 2))
    batch = make_batch(step)  # (x, y_gt, s_gt[, {d_in, y_in}])
    x, y_gt, s_gt = batch[:3]
    feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True, 'generator': gen}
    if len(batch) > 3:
      feed.update(batch[3])
    loss, _ = model.run(['loss', 'train_step'], feed)
    if getattr(args, 'save_rank_weights', False):
      losses.append(float(loss))
    if rank == 0 and (step % args.steps_per_log == 0 or step == args.num_steps - 1):
      print('step %d  loss %.5f  learn_rate %.2e  %.2f s' % (step, float(loss), ra_train.learn_rate(model_opt, step),
                                                             time.time() - t0))
    if rank == 0 and args.save_ckpt and (step + 1) % args.steps_per_ckpt == 0:
      save_checkpoint(ckpt, model)
  model.trainer.flush_status()  # every rank: the last step's solver / controller statuses are checked one step late
  if getattr(args, 'save_rank_weights', False):
    os.makedirs(folder, exist_ok=True)
    np.savez(os.path.join(folder, 'weights_rank%d.npz' % rank), loss_history=np.asarray(losses, np.float64),
             ranks_in_communicator=np.asarray(ra_dist.comm_size()), **model.state_dict_numpy())
  ra_dist.barrier()
  if rank == 0:
    os.makedirs(folder, exist_ok=True)
    if not os.path.exists(os.path.join(folder, 'model_opt.yaml')):
      with open(os.path.join(folder, 'model_opt.yaml'), 'w') as f:
        yaml.safe_dump(model_opt, f)
    save_checkpoint(ckpt, model)
    print('trained %d steps, weights -> %s (weights.npz: weights, EMA shadows, optimizer state)' % (args.num_steps, folder))


def main(argv=None):
  import torch
  args = build_parser().parse_args(argv)
  model_opt = cap.make_model_opt(args, args.inp_height, args.inp_width, args.timespan)
  model_opt['sync_bn'], model_opt['seed'] = bool(args.sync_bn), int(args.seed)
  rank, world, local_rank = ra_dist.init()
  if torch.cuda.is_available():
    torch.cuda.set_device(local_rank)
  model = full_model.get_model(model_opt, is_training=True)
  folder = os.path.join(args.results, args.model_id or 'full_model')
  H, W, T = model_opt['inp_height'], model_opt['inp_width'], model_opt['timespan']
  lo, hi = ra_dist.shard_range(rank, world, args.batch_size)
  if hi <= lo:
    raise SystemExit('batch_size %d < world size %d' % (args.batch_size, world))
  data = dict(np.load(args.input)) if args.input else None

  add_d = bool(model_opt.get('add_d_out', False))  # full_model.py:165-194: d_in and y_in, both or neither
  nsc = int(model_opt.get('num_semantic_classes', 1))

  def make_batch(step):
    if data is None:
      rng = np.random.RandomState(step_seed(args.seed, rank, step))
      b = synthetic_batch(rng, hi - lo, H, W, T)
      return b + (dict(zip(('d_in', 'y_in'), synthetic_extras(rng, hi - lo, H, W, nsc))),) if add_d else b
    idx = (step * args.batch_size + np.arange(lo, hi)) % data['x'].shape[0]
    b = (data['x'][idx], data['y_gt'][idx], data['s_gt'][idx])
    return b + ({'d_in': data['d_in'][idx], 'y_in': data['y_in'][idx]},) if add_d else b

  train_loop(args, model, model_opt, folder, rank, world, make_batch)


if __name__ == '__main__':
  main()
