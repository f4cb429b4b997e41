#!/usr/bin/env python
"""Time the training step's conv launches at one shape: conv_shape_bench.py Cin Cout H W B [reps]
prints us per launch for the float32 conv with batch moments (ra_conv3x3_moments_f32), its bf16-operand form and the
bf16-storage form (ra_conv3x3_bf16_f32, flags 3), each with the algorithmic HBM rate.  RA_CONV_GEO=<gx><gy> forces a tile."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import numpy as np
import torch

import ra_native as rn
if os.environ.get('RA_LIB'):  # A/B: another build of the library (same ABI)
  rn.LIB_PATH = os.environ['RA_LIB']
import ra_ops as ops

cin, cout, H, W, B = [int(v) for v in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 50
dev = torch.device('cuda:0')
rng = np.random.RandomState(0)
x = torch.tensor(rng.randn(B, H, W, cin).astype(np.float32), device=dev)
xb = x.to(torch.bfloat16)
w = (rng.randn(3, 3, cin, cout) * 0.2).astype(np.float32)
wp = torch.tensor(ops.pack_conv_weights(w), device=dev)
cp = ops.cout_padded(cout)
sc, sh = torch.ones(cp, device=dev), torch.zeros(cp, device=dev)
lib = rn.lib()
npf = lib.ra_conv3x3_moments_part_floats(cout)
part, n0 = torch.empty(npf, device=dev), C.c_int(0)
y32 = torch.empty((B, H, W, cout), device=dev)
y16 = torch.empty((B, H, W, cout), dtype=torch.bfloat16, device=dev)


def timeit(fn):
  for _ in range(5):
    fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) * 1e3 / reps


st = rn.stream_ptr()
cases = [
    ('f32 + moments', lambda: lib.ra_conv3x3_moments_f32(rn.ptr(x), cin, None, 0, B, H, W, 0, rn.ptr(wp), rn.ptr(sc), rn.ptr(sh), cout, 0, 0,
                                                         rn.ptr(y32), rn.ptr(part), npf, C.byref(n0), st), 4 * cin + 4 * cout),
    ('bf16 operands + moments', lambda: lib.ra_conv3x3_moments_f32(rn.ptr(x), cin, None, 0, B, H, W, 0, rn.ptr(wp), rn.ptr(sc), rn.ptr(sh), cout,
                                                                   0, 1, rn.ptr(y32), rn.ptr(part), npf, C.byref(n0), st), 4 * cin + 4 * cout),
    ('bf16 storage + moments', lambda: lib.ra_conv3x3_bf16_f32(rn.ptr(xb), cin, None, 0, B, H, W, 0, rn.ptr(wp), rn.ptr(sc), rn.ptr(sh), cout, 0,
                                                               1, rn.ptr(y16), rn.ptr(part), npf, C.byref(n0), 3, st), 2 * cin + 2 * cout),
    ('bf16 storage, no moments', lambda: lib.ra_conv3x3_bf16_f32(rn.ptr(xb), cin, None, 0, B, H, W, 0, rn.ptr(wp), rn.ptr(sc), rn.ptr(sh), cout,
                                                                 0, 1, rn.ptr(y16), None, 0, None, 3, st), 2 * cin + 2 * cout),
    ('f32 plain', lambda: lib.ra_conv3x3_f32(rn.ptr(x), cin, None, 0, B, H, W, 0, rn.ptr(wp), rn.ptr(sc), rn.ptr(sh), cout, 0, 1, None, -1,
                                             rn.ptr(y32), st), 4 * cin + 4 * cout),
]
for name, fn, bpp in cases:
  us = timeit(fn)
  gb = B * H * W * bpp / 1e9
  gf = 2.0 * 9 * cin * cout * B * H * W / 1e9
  print('%-26s %8.1f us   %6.2f TB/s algorithmic   %6.1f TFLOP/s   (geo %s)' % (name, us, gb / us * 1e6 / 1e3, gf / us * 1e6 / 1e3,
                                                                                os.environ.get('RA_CONV_GEO', 'auto')))
