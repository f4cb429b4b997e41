#!/usr/bin/env python
"""Does replaying a captured HIP graph of N kernel nodes issue __amd_rocclr_copyBuffer dispatches of its own
(kernel-argument staging)?  Run under `rocprofv3 --kernel-trace --stats`; the graph below holds no copy node."""
import sys
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
x = torch.zeros(1024, device='cuda')
s = torch.cuda.Stream()
with torch.cuda.stream(s):
  for _ in range(3):
    x.add_(1.0)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
  for _ in range(n):
    x.add_(1.0)
for _ in range(reps):
  g.replay()
torch.cuda.synchronize()
print('nodes', n, 'replays', reps, 'x[0]', float(x[0]))
