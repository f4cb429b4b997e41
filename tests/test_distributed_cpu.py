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
