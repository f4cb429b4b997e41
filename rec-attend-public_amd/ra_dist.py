"""Multi-GPU plumbing for the decode loop: one process per GPU (torch.distributed, backend
'nccl' = RCCL on ROCm, 'gloo' on CPU for tests).  The path shards over independent images
(SURVEY.md §8e): rank r decodes images [r*B/N, (r+1)*B/N) with NO data-path collective;
torch.distributed is used only for the barrier, the max-over-ranks timing and an optional
host-side gather of the small score tensor."""
import os

import torch

import ra_native  # noqa: F401  (sets GPU_MAX_HW_QUEUES before the first HIP call)


def init(backend=None):
  """Initialise from the torchrun environment (RANK / WORLD_SIZE / MASTER_*).  Returns
  (rank, world, local_rank); a no-op single-process setup when WORLD_SIZE is unset or 1."""
  rank = int(os.environ.get('RANK', '0'))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world > 1 and not torch.distributed.is_initialized():
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29511')
    # RA_DIST_BACKEND overrides the caller's choice: 'gloo' moves CUDA tensors through the host, which lets several ranks
    # share ONE GPU (tests/test_distributed_gpu.py: the product's collectives on device tensors where no multi-GPU box exists)
    backend = os.environ.get('RA_DIST_BACKEND') or backend
    if backend is None:
      backend = 'nccl' if torch.cuda.is_available() else 'gloo'
    torch.distributed.init_process_group(backend, rank=rank, world_size=world)
  return rank, world, local_rank


def comm_size():
  """Ranks the initialised communicator (RCCL / gloo) actually holds; 1 without one."""
  if torch.distributed.is_available() and torch.distributed.is_initialized():
    return int(torch.distributed.get_world_size())
  return 1


def shard_range(rank, world, n):
  """Images [lo, hi) of a global batch of n that `rank` decodes: contiguous, sizes differ by
  at most one, every image owned exactly once."""
  base, rem = divmod(n, world)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def barrier():
  if torch.cuda.is_available():
    torch.cuda.synchronize()
  if torch.distributed.is_available() and torch.distributed.is_initialized():
    torch.distributed.barrier()
  if torch.cuda.is_available():
    torch.cuda.synchronize()


def max_over_ranks(seconds):
  """The slowest rank's elapsed time (what whole-job throughput is computed from)."""
  if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
    return float(seconds)
  dev = 'cuda' if torch.distributed.get_backend() == 'nccl' else 'cpu'
  t = torch.tensor([seconds], dtype=torch.float64, device=dev)
  torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
  return float(t.item())


def gather_scores(s_local, world):
  """All ranks' s_out [B_r, T] -> list of tensors (rank order), for host-side assembly."""
  if world == 1:
    return [s_local]
  sizes = [None] * world
  torch.distributed.all_gather_object(sizes, tuple(s_local.shape))
  out = [torch.empty(sz, dtype=s_local.dtype, device=s_local.device) for sz in sizes]
  torch.distributed.all_gather(out, s_local.contiguous()) if len(set(sizes)) == 1 else \
      torch.distributed.all_gather_object(out, s_local.cpu())
  return out
