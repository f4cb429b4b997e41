#!/usr/bin/env python
"""Benchmark of the recurrent-attention decode loop on MI355X.

One "step" = one eval-mode full_model forward (all T instance timesteps) of a batch of B
synthetic CVPPP-shaped images per GPU: the workload BASELINE.json's metric is quoted on
(configs[1]: CVPPP arch, 512x512, T=16, B=8).  Inputs and weights are resident in HBM before
the timed region.  Multi-GPU: one process per GPU, every rank decodes its own shard of B
images (the path is embarrassingly parallel over the batch; no data-path collective), so
scaling is weak.

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel group — the controller
CNN's conv3x3 f32-MFMA launches (1.585 GFLOP per image-timestep at cfg2) —
timed with HIP events on the launch stream; `roofline_attn` the HBM-bound attention resample
(extract + paste, 7.0 MiB algorithmic per image-timestep); `cpu_baseline` the NumPy oracle
timed on this box's host cores over a bounded sample (a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd'))

import numpy as np
import torch

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, exact f32
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec (6.29 TB/s measured copy)


def make_opt(arch, H, W, T):
  """model_opt of the reference's run scripts (run_cvppp.sh:37-72 / run_kitti.sh:68-111 via
  full_model_train.py:581-658)."""
  opt = dict(
      inp_height=H, inp_width=W, inp_depth=3, padding=16, filter_height=48, filter_width=48,
      timespan=T, ctrl_rnn_hid_dim=256, num_ctrl_mlp_layers=1, ctrl_mlp_dim=256,
      mlp_dropout=None, weight_decay=5e-5, use_bn=True, attn_box_padding_ratio=0.2,
      use_knob=False, squash_ctrl_params=False, fixed_order=False, fixed_gamma=True,
      fixed_var=False, dynamic_var=False, num_ctrl_rnn_iter=5, num_glimpse_mlp_layers=2,
      stop_canvas_grad=True, use_iou_box=False, add_skip_conn=False, disable_overwrite=False,
      add_d_out=False, add_y_out=False, num_semantic_classes=1, ctrl_add_inp=True,
      ctrl_add_canvas=True, ctrl_add_d_out=False, ctrl_add_y_out=False, attn_add_inp=True,
      attn_add_canvas=True, attn_add_d_out=False, attn_add_y_out=False,
      ctrl_cnn_filter_size=[3] * 8, ctrl_cnn_depth=[8, 8, 16, 16, 32, 32, 64, 64],
      ctrl_cnn_pool=[1, 2, 1, 2, 1, 2, 2, 2], attn_cnn_filter_size=[3] * 6,
      attn_cnn_depth=[8, 8, 16, 16, 32, 32], attn_cnn_pool=[1, 2, 1, 2, 1, 2],
      attn_dcnn_filter_size=[3] * 7, attn_dcnn_depth=[32, 32, 16, 16, 8, 8, 1],
      attn_dcnn_pool=[2, 1, 2, 1, 2, 1, 1], attn_cnn_skip='1,1,1')
  if arch != 'cvppp':
    raise ValueError(arch)
  return opt


def encoder_flops_per_image(d):
  """sum over ctrl-CNN layers of 2*9*Cin*Cout*Hc*Wc."""
  h, w, tot, per = d['H'], d['W'], 0.0, []
  for i in range(d['ccnn_nlayers']):
    f = 2.0 * 9 * d['ccnn_channels'][i] * d['ccnn_channels'][i + 1] * h * w
    per.append(f)
    tot += f
    h, w = h // d['ccnn_pool'][i], w // d['ccnn_pool'][i]
  return tot, per


def seed_weights(model, seed):
  """Synthetic weights: the reference's truncated_normal(0.01) init is already in place
  (nnlib.py:53-54); give the BN EMA statistics random positive values and
  a plausible box so the Gaussian bands have realistic extent.  Values do not affect speed."""
  g = torch.Generator().manual_seed(seed)
  for k in model.weight_keys():
    t = model[k]
    if k.endswith('_ema_var'):
      t.copy_(torch.empty(t.shape).uniform_(0.5, 1.5, generator=g))
    elif k.endswith('_ema_mean'):
      t.copy_(torch.empty(t.shape).normal_(0.0, 0.05, generator=g))
  b = model['ctrl_mlp_b_0'].cpu()
  b[2:4] = float(np.log(0.35))
  model['ctrl_mlp_b_0'].copy_(b)


def cpu_baseline(opt, seed, budget_s=20.0):
  """NumPy oracle (oracle/ra_oracle.py, float32) on the host cores: B=1, as many timesteps
  as fit the budget (each timestep costs the same)."""
  sys.path.insert(0, os.path.join(ROOT, 'oracle'))
  import ra_oracle as ora
  try:
    from threadpoolctl import threadpool_info
    threads = max([p.get('num_threads', 1) for p in threadpool_info()] or [1])
  except Exception:
    threads = os.cpu_count() or 1
  o1 = dict(opt)
  o1['timespan'] = 1
  P = ora.random_params(o1, seed)
  x = np.random.RandomState(seed).rand(1, opt['inp_height'], opt['inp_width'], 3).astype(np.float32)
  ora.full_model_forward(o1, P, x, dtype=np.float32)  # warm-up
  n, t0 = 0, time.perf_counter()
  while True:
    ora.full_model_forward(o1, P, x, dtype=np.float32)
    n += 1
    el = time.perf_counter() - t0
    if el > budget_s or n >= 16:
      break
  return {'value': n / el, 'unit': 'instance-timesteps/s', 'cores': int(threads),
          'kind': 'port',
          'sample': 'NumPy/BLAS float32 oracle, %d x (B=1, T=1) forwards of the same %dx%d '
                    'CVPPP-arch graph in %.1f s; stand-in for the TF-0.12 CPU path, which cannot '
                    'run here' % (n, opt['inp_height'], opt['inp_width'], el)}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=20)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--batch', type=int, default=8, help='images per GPU')
  ap.add_argument('--size', type=int, default=512)
  ap.add_argument('--timespan', type=int, default=16)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-graph', action='store_true')
  ap.add_argument('--profile-steps', type=int, default=3,
                  help='instrumented (HIP-event per stage) forwards for the roofline objects')
  args = ap.parse_args()

  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      raise SystemExit('launch with torch.distributed.run --nproc-per-node %d' % args.gpus)
  torch.cuda.set_device(local_rank)
  dist = None
  if world > 1:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('nccl', rank=rank, world_size=world)

  import full_model
  B, T, S = args.batch, args.timespan, args.size
  opt = make_opt('cvppp', S, S, T)
  model = full_model.get_model(opt, is_training=False)
  seed_weights(model, 1234 + rank)
  eng = model.engine
  eng.use_graph = not args.no_graph
  g = torch.Generator().manual_seed(1234 + rank)
  x = torch.rand((B, S, S, 3), generator=g, dtype=torch.float32).cuda()
  feed = {'x': x, 'phase_train': False}

  def barrier():
    torch.cuda.synchronize()
    if dist is not None:
      dist.barrier()
    torch.cuda.synchronize()

  for _ in range(max(args.warmup, 1)):
    eng.forward(feed['x'])
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    eng.forward(feed['x'])
  barrier()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    tmax = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
  value = world * B * T * args.steps / elapsed

  out = {
      'metric': 'instance-timesteps/sec, full_model forward 512x512 T=16 (whole job)',
      'value': value, 'unit': 'instance-timesteps/s', 'per_gpu': value / world,
      'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
      'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
      'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'config': {'workload': 'cfg2: CVPPP-arch full_model eval forward, %dx%d, T=%d, B=%d per GPU, '
                             'y_out+s_out' % (S, S, T, B),
                 'arch': 'cvppp', 'H': S, 'W': S, 'T': T, 'batch_per_gpu': B,
                 'global_batch': B * world, 'parallelism': 'batch-sharded x%d, no collective' % world,
                 'hip_graph': bool(eng.use_graph)},
  }

  if rank == 0:
    # ---- roofline objects: instrumented forwards (HIP events between stages, same stream) ----
    eng.use_graph = False
    acc = {}
    for _ in range(max(args.profile_steps, 1)):
      eng.timing = []
      eng._mark('start')
      eng.forward(feed['x'])
      torch.cuda.synchronize()
      for k, (ms, n) in eng.stage_times_ms().items():
        a = acc.setdefault(k, [0.0, 0])
        a[0] += ms
        a[1] += n
    eng.timing = None
    d = model.dims
    tot_f, per_f = encoder_flops_per_image(d)
    layers = []
    enc_ms_per_ts = 0.0
    for i in range(d['ccnn_nlayers']):
      ms, n = acc['ctrl_cnn_L%d' % i]
      avg = ms / n
      enc_ms_per_ts += avg
      layers.append({'layer': i, 'avg_us': 1e3 * avg, 'gflop': per_f[i] * B / 1e9,
                     'tflops': per_f[i] * B / (avg * 1e-3) / 1e12})
    achieved = tot_f * B / (enc_ms_per_ts * 1e-3) / 1e12
    out['roofline'] = {
        'kernel': 'ra::conv::conv3x3_mfma (controller CNN, %d launches per timestep)' % d['ccnn_nlayers'],
        'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
        'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': None,
        'peak_note': 'dense f32-input MFMA (v_mfma_f32_16x16x4_f32); the kernel computes in exact f32',
        'flop_per_launch_group': tot_f * B, 'avg_us_per_launch_group': 1e3 * enc_ms_per_ts,
        'layers': layers}
    attn_ms = sum(acc[k][0] / acc[k][1] for k in ('extract', 'paste') if k in acc)
    attn_bytes = float(S * S * (d['acnn_channels'][0] + 3) * 4) * B
    out['roofline_attn'] = {
        'kernel': 'extract_patch + paste_u + paste (attention resample)', 'bound': 'hbm',
        'achieved': attn_bytes / (attn_ms * 1e-3) / 1e9, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
        'frac': attn_bytes / (attn_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 'traffic': None,
        'bytes_per_launch_group': attn_bytes, 'avg_us_per_launch_group': 1e3 * attn_ms}
    out['stage_us_per_timestep'] = {k: 1e3 * v[0] / v[1] for k, v in sorted(acc.items())}
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline(opt, 1234)
    print(json.dumps(out))
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
