"""The N > 1 path on CPU: two gloo ranks shard a global batch (no data-path collective), run the
barrier / max-over-ranks timing used by bench.py and gather the per-rank scores."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

import ra_dist


def test_shard_range_partitions():
  for n in (1, 7, 8, 32, 33):
    for world in (1, 2, 3, 8):
      spans = [ra_dist.shard_range(r, world, n) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
      sizes = [hi - lo for lo, hi in spans]
      assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  sys.path.insert(0, os.path.dirname(ra_dist.__file__))
  r, w, _ = ra_dist.init('gloo')
  lo, hi = ra_dist.shard_range(r, w, 8)
  # stand-in for the rank's decode: a deterministic function of its own images only
  x = torch.arange(8, dtype=torch.float32)[lo:hi]
  s_local = torch.stack([x, x * 2], dim=1)
  ra_dist.barrier()
  t = ra_dist.max_over_ranks(0.5 + 0.25 * r)
  parts = ra_dist.gather_scores(s_local, w)
  ra_dist.barrier()
  q.put((r, lo, hi, t, torch.cat(parts).numpy()))
  torch.distributed.destroy_process_group()


def test_two_rank_gloo():
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29500 + (os.getpid() % 400)
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = sorted(q.get(timeout=120) for _ in procs)
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert [(r[1], r[2]) for r in res] == [(0, 4), (4, 8)]
  assert all(abs(r[3] - 0.75) < 1e-12 for r in res)           # max over ranks
  full = np.stack([np.arange(8), 2 * np.arange(8)], 1).astype(np.float32)
  assert all((r[4] == full).all() for r in res)               # every image decoded exactly once


# ---------------------------------------------------------------------------------------------
# Data-parallel training step: ONE flat bucket, ONE all-reduce, then
# scale -> weight decay -> clip -> Adam.  The collective and the bucket bookkeeping are the
# product code (ra_train.GradBucket, gloo here, RCCL on the GPUs); the optimizer arithmetic the
# HIP kernel performs is restated in NumPy below (its own parity test runs on the GPU).
class _FakeModel(dict):
  def __init__(self, tensors, opt):
    dict.__init__(self, tensors)
    self.opt = opt

  def weight_keys(self):
    return sorted(k for k, v in self.items() if isinstance(v, torch.Tensor))


def _toy_model():
  g = torch.Generator().manual_seed(7)
  shapes = {'ctrl_cnn_w_0': (3, 3, 4, 8), 'ctrl_cnn_b_0': (8,), 'ctrl_cnn_0_0_beta': (8,), 'ctrl_cnn_0_0_gamma': (8,),
            'ctrl_cnn_0_0_ema_mean': (8,), 'ctrl_lstm_w_xi': (5, 7), 'score_mlp_w_0': (9, 1)}
  opt = dict(weight_decay=5e-5, base_learn_rate=1e-3, learn_rate_decay=0.96, steps_per_learn_rate_decay=5000)
  return _FakeModel({k: torch.randn(s, generator=g) for k, s in shapes.items()}, opt)


def _per_example_grads(n_total):
  """Deterministic per-example gradients (the same on every rank), several beyond the +-1 clip."""
  import ra_train
  m = _toy_model()
  b = ra_train.GradBucket(m)
  g = torch.Generator().manual_seed(11)
  return 4.0 * torch.randn((n_total, b.n), generator=g)


def _reference_update(param, grad_mean, wd, lr_t):
  gg = np.clip(grad_mean + wd * param, -1.0, 1.0)
  m = 0.1 * gg
  v = 0.001 * gg * gg
  return param - lr_t * m / (np.sqrt(v) + 1e-7)


def _train_worker(rank, world, port, q):
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  sys.path.insert(0, os.path.dirname(ra_dist.__file__))
  import ra_train
  r, w, _ = ra_dist.init('gloo')
  model = _toy_model()
  bucket = ra_train.GradBucket(model)
  per_ex = _per_example_grads(8)
  lo, hi = ra_dist.shard_range(r, w, 8)
  # the rank's backward: its loss divides by ITS example count (full_model.py:916 on the shard)
  bucket.grad.copy_(per_ex[lo:hi].mean(dim=0))
  n = bucket.allreduce()
  q.put((r, n, bucket.grad.numpy().copy(), bucket.param.numpy().copy(), bucket.wd.numpy().copy(),
         {k: bucket.offsets[k] for k in bucket.names}))
  ra_dist.barrier()
  torch.distributed.destroy_process_group()


def test_two_rank_gradient_bucket_equals_single_process():
  import ra_train
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29950 + (os.getpid() % 40)
  procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  single = _per_example_grads(8).mean(dim=0).numpy()       # one process, the whole batch
  lr_t = ra_train.learn_rate(_toy_model().opt, 0) * np.sqrt(1 - 0.999) / (1 - 0.9)
  for r, world, gsum, param, wd, offsets in res:
    assert world == 2
    assert (gsum == res[0][2]).all()                        # every rank holds the same sum
    assert np.abs(gsum / world - single).max() < 1e-6       # sum-then-scale == single-process mean
    upd_dp = _reference_update(param, gsum / world, wd, lr_t)
    upd_single = _reference_update(param, single, wd, lr_t)
    assert np.abs(upd_dp - upd_single).max() < 1e-7         # ... and so is the clipped Adam update
    # clipping BEFORE the mean would differ: the order matters and is the documented one
    assert np.abs(np.clip(gsum, -1, 1) / world - np.clip(single, -1, 1)).max() > 1e-3
    # bucket layout: EMA shadows are not trainable, weight decay only on `w` tensors
    assert 'ctrl_cnn_0_0_ema_mean' not in offsets
    o, n, _ = offsets['ctrl_cnn_w_0']
    assert (wd[o:o + n] == np.float32(5e-5)).all()
    o, n, _ = offsets['ctrl_cnn_b_0']
    assert (wd[o:o + n] == 0).all()


def test_learn_rate_and_knob_schedules():
  import ra_train
  opt = dict(base_learn_rate=1e-3, learn_rate_decay=0.96, steps_per_learn_rate_decay=5000, knob_base=1.0,
             knob_decay=0.9, steps_per_knob_decay=300)
  assert ra_train.learn_rate(opt, 4999) == 1e-3 and abs(ra_train.learn_rate(opt, 5000) - 0.96e-3) < 1e-12
  assert ra_train.knob_prob(opt, 100, 200) == 1.0 and abs(ra_train.knob_prob(opt, 500, 200) - 0.9) < 1e-12


# ---------------------------------------------------------------------------------------------
# every process draws its own initial weights; a data-parallel trainer must start
# from ONE model.  Two gloo ranks build the model through get_model() with different torch seeds,
# wrap it in the product's GradBucket, and take rank 0's state; also the whole-batch BatchNorm
# moments (nnlib.py:98) from per-shard moments with the product's gather + Chan combination, and the
# optimizer-state round trip of the checkpoint.
def _state_worker(rank, world, port, q):
  os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  here = os.path.dirname(ra_dist.__file__)
  sys.path.insert(0, here)
  sys.path.insert(0, os.path.join(os.path.dirname(here), 'oracle'))
  import full_model
  import ra_oracle as ora
  import ra_train
  r, w, _ = ra_dist.init('gloo')
  torch.manual_seed(100 + 17 * r)  # what separate processes do anyway: different initial draws
  opt = ora.make_opt('cvppp', 64, 64, 2)
  model = full_model.get_model(opt)
  bucket = ra_train.GradBucket(model)
  before = bucket.param.clone()
  bucket.m.fill_(float(r + 1))
  bucket.global_step = 5 * r + 2
  n = bucket.broadcast(0)
  # whole-batch moments from the two shards' moments
  g = torch.Generator().manual_seed(3)
  xall = torch.randn((8, 6, 5, 4), generator=g) * 3 + 1.5
  lo, hi = ra_dist.shard_range(r, w, 8)
  xs = xall[lo:hi].reshape(-1, 4)
  mean, var = xs.mean(dim=0), xs.var(dim=0, unbiased=False)
  ntot = ra_train.sync_moments(mean, var, xs.shape[0])
  q.put((r, n, before.numpy(), bucket.param.numpy().copy(), bucket.m.numpy().copy(), bucket.global_step,
         float(model['global_step']), mean.numpy(), var.numpy(), ntot, model['ctrl_cnn_w_0'].numpy().copy()))
  ra_dist.barrier()
  torch.distributed.destroy_process_group()


def test_two_rank_initial_state_is_rank0s_and_moments_are_whole_batch():
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29700 + (os.getpid() % 200)
  procs = [ctx.Process(target=_state_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  res = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
  for p in procs:
    p.join(timeout=60)
    assert p.exitcode == 0
  (r0, n0, b0, p0, m0, gs0, mg0, mean0, var0, nt0, w0), (r1, n1, b1, p1, m1, gs1, mg1, mean1, var1, nt1, w1) = res
  assert n0 == n1 == 2
  assert np.abs(b0 - b1).max() > 1e-3                  # the ranks' own draws differed ...
  assert (p0 == b0).all() and (p1 == b0).all()         # ... and both now hold rank 0's weights,
  assert (w1 == w0).all()                              # through the model's views as well,
  assert (m0 == 1.0).all() and (m1 == 1.0).all()       # its Adam moments
  assert gs0 == gs1 == 2 and mg0 == mg1 == 2.0         # and its global_step (LR staircase, knob schedules)
  g = torch.Generator().manual_seed(3)
  xall = (torch.randn((8, 6, 5, 4), generator=g) * 3 + 1.5).reshape(-1, 4)
  for mean, var, nt in ((mean0, var0, nt0), (mean1, var1, nt1)):
    assert nt == xall.shape[0]
    assert np.abs(mean - xall.mean(dim=0).numpy()).max() < 1e-6
    assert np.abs(var - xall.var(dim=0, unbiased=False).numpy()).max() < 1e-5


def test_optimizer_state_round_trip_and_combine_moments():
  import ra_train
  a, b = ra_train.GradBucket(_toy_model()), ra_train.GradBucket(_toy_model())
  g = torch.Generator().manual_seed(5)
  a.m.copy_(torch.randn(a.n, generator=g))
  a.v.copy_(torch.rand(a.n, generator=g))
  a.global_step = 7
  st = a.state_dict()
  assert set(k for k in st if k != 'global_step') == {s + k for s in ('adam_m/', 'adam_v/') for k in a.names}
  b.load_state_dict(st)
  for k in a.names:
    o, n, _ = a.offsets[k]
    assert (b.m[o:o + n] == a.m[o:o + n]).all() and (b.v[o:o + n] == a.v[o:o + n]).all()
  assert b.global_step == 7 and b.model['global_step'] == 7.0
  import pytest
  with pytest.raises(Exception):
    b.load_state_dict({'global_step': np.asarray(1)})    # strict: missing Adam slots
  # unequal shard sizes: Chan's combination is exact for any split
  x = torch.randn((37, 3), generator=g, dtype=torch.float64) * 2 + 5
  parts = [x[:5], x[5:30], x[30:]]
  n, m, v = ra_train.combine_moments(torch.tensor([float(p.shape[0]) for p in parts], dtype=torch.float64),
                                     torch.stack([p.mean(dim=0) for p in parts]),
                                     torch.stack([p.var(dim=0, unbiased=False) for p in parts]))
  assert float(n) == 37 and (m - x.mean(dim=0)).abs().max() < 1e-12 and (v - x.var(dim=0, unbiased=False)).abs().max() < 1e-12
