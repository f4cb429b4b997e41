#!/usr/bin/env python
"""The filter-gradient kernel at one layer shape: python tools/wgrad_shape_bench.py Cin Cout H W B  (float32; RA_LIB = another
build of the library for A/B runs, RA_WGRAD8=0 / RA_WGRAD_SMALL=0 pick the older forms)."""
import ctypes as C
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, os.path.join(ROOT, 'rec-attend-public_amd'))
import torch

import ra_native as rn
if os.environ.get('RA_LIB'):
  rn.LIB_PATH = os.environ['RA_LIB']

cin, cout, H, W, B = (int(v) for v in sys.argv[1:6])
dev = torch.device('cuda:0')
x = torch.randn(B, H, W, cin, device=dev)
du = torch.randn(B, H, W, cout, device=dev)
lib = rn.lib()
n = lib.ra_conv3x3_wgrad_workspace_floats(cin, cout, B, H, W)
ws = torch.empty(n, device=dev)
dw, db = torch.empty(3, 3, cin, cout, device=dev), torch.empty(cout, device=dev)
run = lambda: rn.check(lib.ra_conv3x3_wgrad_f32(rn.ptr(x), cin, B, H, W, 0, rn.ptr(du), cout, rn.ptr(ws), n, rn.ptr(dw), rn.ptr(db),
                                                 rn.stream_ptr()), 'wgrad')
for _ in range(3):
  run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
  run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100.0
gb = B * H * W * (cin + cout) * 4 / 1e9
print('wgrad %d -> %d, %d x %d x %d: %.1f us per call (kernel + final reduction), %.2f TB/s algorithmic' % (cin, cout, H, W, B, us, gb / (us * 1e-6) * 1e-3))
