"""Time ra_hungarian_f32_dev on f_segm_match-shaped problems (modellib.py:395-411): weights =
iou * mask + 1e-5 with the first k ground-truth columns live.  Prints µs per call for a few mixes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'rec-attend-public_amd'))
import ra_ops as ops  # noqa: E402


def problems(B, T, live, seed):
  rng = np.random.RandomState(seed)
  iou = rng.uniform(0, 1, (B, T, T)).astype(np.float32)
  mask = np.zeros((B, 1, T), np.float32)
  for b in range(B):
    k = live if live > 0 else rng.randint(1, T)
    mask[b, 0, :k] = 1
  return torch.from_numpy(iou * mask + 1e-5).cuda()


def main():
  B, T = 8, 21
  for live in (0, 5, 12, 20, 21):
    w = problems(B, T, live, 7)
    for _ in range(2):
      ops.hungarian(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
      m = ops.hungarian(w)[0]
    e1.record()
    torch.cuda.synchronize()
    cpu = ops.hungarian(w.cpu())[0]
    same = bool((m.cpu() == cpu).all())
    print('T=%d live=%s: %.1f us/call, device == host: %s' % (T, live or 'mixed', e0.elapsed_time(e1) * 1e3 / n, same))


if __name__ == '__main__':
  main()
