#!/usr/bin/env python
"""us per launch and HBM rate of the training step's float4 BatchNorm kernels at one layer shape (float32):
ra_bn_act_pool_f32 (forward), ra_bn_act_pool_bwd_f32 (reduce + finish + dx).  usage: bn_shape_bench.py C H W B pool"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import torch
import ra_native as rn
if os.environ.get('RA_LIB'):  # A/B: another build of the library (same ABI)
  rn.LIB_PATH = os.environ['RA_LIB']
Cc, H, W, B, pool = [int(v) for v in sys.argv[1:6]]
dev = torch.device('cuda')
L = rn.lib()
u = torch.randn(B, H, W, Cc, device=dev)
dy = torch.randn(B, H // pool, W // pool, Cc, device=dev)
y = torch.empty_like(dy)
du = torch.empty_like(u)
mean, var = u.mean(dim=(0, 1, 2)).contiguous(), u.var(dim=(0, 1, 2), unbiased=False).contiguous()
gamma, beta = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
ws = torch.empty(L.ra_bn_workspace_floats(Cc), device=dev)
dg, db = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
p = rn.ptr
st = lambda: rn.stream_ptr()
fwd = lambda: rn.check(L.ra_bn_act_pool_f32(p(u), p(mean), p(var), p(gamma), p(beta), C.c_float(1e-3), 1, pool, B, H, W, Cc, p(y), st()), 'fwd')
bwd = lambda: rn.check(L.ra_bn_act_pool_bwd_f32(p(u), p(dy), p(mean), p(var), p(gamma), p(beta), C.c_float(1e-3), 1, pool, B, H, W, Cc,
                                                p(ws), ws.numel(), p(dg), p(db), p(du), st()), 'bwd')


def graph_us(fn, inner=4, reps=10):
  fn(); torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(inner):
      fn()
  g.replay()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  e0.record()
  for _ in range(reps):
    g.replay()
  e1.record()
  torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / (reps * inner)


nu, ny = u.numel() * 4, dy.numel() * 4
t = graph_us(fwd)
print('C %d %dx%d B %d pool %d: forward  %7.1f us  %.2f TB/s (reads u, writes y)' % (Cc, H, W, B, pool, t, (nu + ny) / t / 1e6))
t = graph_us(bwd)
print('%38s backward %7.1f us  %.2f TB/s (reduce: u + dy; dx: u + dy + du)' % ('', t, (3 * nu + 2 * ny) / t / 1e6))
