#!/usr/bin/env python
"""Entry point with the flag surface of the reference's full_model_eval.py (:180-222).

Restores results/<model_id>/{model_opt.yaml, weights.npz}, forces `use_knob=False`
(full_model_eval.py:172-174), feeds {x, phase_train=False[, d_in, y_in]} and fetches
['y_out', 's_out'] (full_model_eval.py:35; runner.py:91-105) batch by batch on the MI355X
kernels, sharding the images over the ranks when launched with torch.distributed.run.  Inputs
come from --input (an .npz with x [N,H,W,3] and optionally d_in / y_in / y_gt / s_gt / fg) or are
synthetic; the reference's HDF5 datasets are out of scope.  The raw outputs are
written to <output>/output_<split>/pred_rank<r>.npz.  With y_gt and s_gt in the input the
reference's write_log chain (full_model_eval.py:97-139) runs on the device for every threshold of
--threshold_list (default 0.3, the reference CLI's default, :193-194): apply_confidence, apply_one_label,
apply_threshold [, mask_foreground, remove_tiny] and the --analyzers (default list :201-205);
the per-threshold means go to <output>/output_<split>/metrics_rank<r>.yaml.  The cv2 steps
(upsample + bilateral filter, morph) run as device kernels (utils/postprocess.py).  As in the reference, pp.upsample — cv2.resize
to the labels' size, then bilateralFilter(5, 10, 10), which on [0, 1] maps is close to a radius-2 blur — ALWAYS runs, also when
the labels already have the network's size (full_model_eval.py:114), and the dilation runs when a foreground mask is given and
--no_morph is not.  --fused_postprocess (not a reference flag) skips the bilateral step when no resize is needed: the chain
then collapses into one fused device pass; masks and metrics near the threshold differ from the reference's in that mode."""
import argparse
import os
import time

import numpy as np
import yaml

import cmd_args_parser as cap
import full_model
import ra_dist


def build_parser():
  p = argparse.ArgumentParser(description='Evaluate output')
  for table in (cap.EVAL_FLAGS, cap.DATA_FLAGS):
    cap.add_flags(p, table)
  p.add_argument('--input', default=None, help='.npz with x [N,H,W,3] (+ d_in, y_in)')
  p.add_argument('--num_synthetic', type=int, default=8)
  p.add_argument('--fused_postprocess', action='store_true',
                 help='not a reference flag: skip upsample\'s bilateral filter when the labels have the network\'s size (one fused '
                      'post-processing pass; results near the threshold differ from the reference chain)')
  p.add_argument('--in_flight', type=int, default=8, help='batches submitted to one GPU at a time (four decode concurrently, the rest queue behind them)')
  return p


def main(argv=None):
  import torch
  args = build_parser().parse_args(argv)
  if args.model_id is None:
    raise Exception('You must provide model ID')  # cmd_args_parser.py:154-155
  restore = os.path.join(args.results, args.model_id)
  with open(os.path.join(restore, 'model_opt.yaml')) as f:
    model_opt = yaml.safe_load(f)
  model_opt['use_knob'] = False
  rank, world, local_rank = ra_dist.init()
  if torch.cuda.is_available():
    torch.cuda.set_device(local_rank)
  model = full_model.get_model(model_opt).load_weights(dict(np.load(os.path.join(restore, 'weights.npz'))))
  H, W = model_opt['inp_height'], model_opt['inp_width']
  if args.input:
    data = dict(np.load(args.input))
  else:
    rng = np.random.RandomState(1234)
    data = {'x': rng.rand(args.num_synthetic, H, W, 3).astype(np.float32)}
    if model.dims['add_d_out']:
      n = args.num_synthetic
      data['d_in'] = np.eye(8, dtype=np.float32)[rng.randint(0, 8, (n, H, W))]
      lg = rng.randn(n, H, W, model.dims['nsc']).astype(np.float32)
      data['y_in'] = np.exp(lg) / np.exp(lg).sum(-1, keepdims=True)
  lo, hi = ra_dist.shard_range(rank, world, data['x'].shape[0])
  # the reference CLI defaults to [0.3] (MyEvalArgsParser.make_opt, full_model_eval.py:193-198); the
  # 0.0 .. 0.9 sweep of :39-40 is only EvalRunner's fallback for a caller that hands it None
  thresholds = [float(t) for t in args.threshold_list.split(',')] if args.threshold_list else [0.3]
  if args.analyzers is None:                                          # :199-210
    names = [] if args.test else ['sbd', 'wt_cov', 'unwt_cov', 'avg_fp', 'avg_fn', 'avg_pr', 'avg_re',
                                  'obj_pr', 'obj_re', 'count_acc', 'count_mse', 'dic', 'dic_abs']
  else:
    names = [n for n in args.analyzers.split(',') if n]
  analyze = bool(names) and 'y_gt' in data and 's_gt' in data
  acc = {th: {n: [] for n in names} for th in thresholds}
  ys, ss, t0 = [], [], time.time()
  try:
    _run_shard(args, model, data, lo, hi, rank, restore, thresholds, names, analyze, acc, ys, ss, t0)
  finally:
    ra_dist.barrier()  # always reached: a rank that fails must not leave the others hanging


def _run_shard(args, model, data, lo, hi, rank, restore, thresholds, names, analyze, acc, ys, ss, t0):
  import torch
  def consume(b0, b1, y_dev, s_dev):
    if analyze:
      import analysis
      from utils import postprocess as pp
      dev = y_dev.device
      gt = torch.as_tensor(np.asarray(data['y_gt'][b0:b1], dtype=np.float32)).to(dev)
      sg = torch.as_tensor(np.asarray(data['s_gt'][b0:b1], dtype=np.float32)).to(dev)
      fg = torch.as_tensor(np.asarray(data['fg'][b0:b1], dtype=np.float32)).to(dev) if 'fg' in data else None
      # the reference's chain (full_model_eval.py:112-124): apply_confidence -> upsample to the labels' size (ALWAYS: resize +
      # bilateral filter, also at equal size) -> [foreground given: morph unless --no_morph] -> apply_one_label -> per threshold:
      # apply_threshold [-> mask_foreground -> remove_tiny].  --fused_postprocess: without a resize and a dilation the chain
      # collapses into ONE fused pass (pp.postprocess) that leaves the bilateral step out
      resize = tuple(gt.shape[-2:]) != tuple(y_dev.shape[-2:]) or not getattr(args, 'fused_postprocess', False)
      dilate = fg is not None and not args.no_morph
      if resize or dilate:
        s2 = s_dev[:, :, 0].contiguous() if s_dev.dim() == 3 else s_dev   # multi-class: :108-110
        yc, s_conf = pp.apply_confidence(y_dev, s2)
        if resize:
          yc = pp.upsample(yc, gt)
        if dilate:
          yc = pp.morph(yc)
        one = pp.apply_one_label(yc)
      for th in thresholds:
        if resize or dilate:
          y_bin, s_hard = pp.apply_threshold(one, th), s_conf
          if fg is not None:
            y_bin, s_hard = pp.remove_tiny(pp.mask_foreground(y_bin, fg), s_hard, threshold=args.remove_tiny)
        else:
          y_bin, s_hard, _ = pp.postprocess(y_dev, s_dev, th, fg=fg, remove_tiny_threshold=args.remove_tiny)
        results = {'y_out': y_bin, 'y_gt': gt, 's_out': s_hard, 's_gt': sg}
        for n in names:
          acc[th][n].append(analysis.create_analyzer(n)(results).cpu().numpy())
    ys.append(y_dev if isinstance(y_dev, np.ndarray) else y_dev.cpu().numpy())  # to_host: already host arrays
    ss.append(s_dev if isinstance(s_dev, np.ndarray) else s_dev.cpu().numpy())

  # batches are independent: keep several in flight, each on its own HIP stream, so that the
  # latency-bound tail of one decodes under the controller CNN of the others (DecodePipeline)
  # results copied to the host ride on the slot's stream: a batch queued behind one would wait for its copy, so that
  # mode keeps one batch per stream (33.8k vs 29.0k instance-timesteps/s at cfg2)
  depth = max(1, args.in_flight) if analyze else max(1, min(args.in_flight, 4))
  # small batches travel together, about 16 images to a slot and forward (DecodePipeline(coalesce=...), round 5: two batches of 8
  # +15 % at cfg2, four batches of 4 +70 % at cfg5 with the same images in flight), as long as at least two slots remain
  coalesce = 1
  if analyze and depth >= 2 and args.batch_size <= 8:
    coalesce = max(1, min(16 // max(1, args.batch_size), depth // 2))
  pipe, spans = model.pipeline(max(1, depth // coalesce), coalesce=coalesce), []
  for b0 in range(lo, hi, args.batch_size):
    b1 = min(hi, b0 + args.batch_size)
    feed = {k: v[b0:b1] for k, v in data.items() if k in ('x', 'd_in', 'y_in')}
    feed['phase_train'] = False
    left = -(-(hi - b1) // args.batch_size)  # batches still to come: the last few go out one per slot (DecodePipeline._ends_soon)
    while pipe.full(b1 - b0, remaining=left):
      consume(*(spans.pop(0) + tuple(pipe.collect())))
    pipe.submit(['y_out', 's_out'], feed, to_host=not analyze, remaining=left)
    spans.append((b0, b1))
  while len(pipe):
    consume(*(spans.pop(0) + tuple(pipe.collect())))
  out_dir = os.path.join(args.output or restore, 'output_' + args.split.split(',')[0])
  os.makedirs(out_dir, exist_ok=True)
  path = os.path.join(out_dir, 'pred_rank%d.npz' % rank)
  T, H, W = model.dims['T'], model.dims['H'], model.dims['W']
  # an empty shard (more ranks than images) writes empty arrays of the right rank
  y_all = np.concatenate(ys) if ys else np.zeros((0, T, H, W), np.float32)
  s_all = np.concatenate(ss) if ss else np.zeros((0, T), np.float32)
  np.savez_compressed(path, y_out=y_all, s_out=s_all, first_index=lo)
  print('rank %d: images [%d, %d) -> %s (%.2f s)' % (rank, lo, hi, path, time.time() - t0))
  if analyze:
    summary = {}
    for th in thresholds:
      vals = {n: np.concatenate(v) if v else np.zeros(0) for n, v in acc[th].items()}
      summary['%.2f' % th] = {n: {'mean': float(v.mean()) if v.size else 0.0, 'count': int(v.size)}
                              for n, v in vals.items()}
    with open(os.path.join(out_dir, 'metrics_rank%d.yaml' % rank), 'w') as f:
      yaml.safe_dump(summary, f)
    for th in sorted(summary):  # every threshold, never a ground-truth-tuned arg-max
      print('rank %d: threshold %s  %s' % (rank, th, ' '.join(
          '%s=%.4f' % (n, summary[th][n]['mean']) for n in names)))


if __name__ == '__main__':
  main()
