#!/usr/bin/env python
"""Ablation timing of the Winograd conv kernel: tools/wv/wino_v{N}.so are builds of csrc/ra_conv_wino.hip with
one part removed (1: filters not loaded, 2: no MFMA loop, 3: no output stores, 4: no input loads)."""
import ctypes as C, glob, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import numpy as np, torch
import ra_ops as ops
def t_us(fn, reps=50):
  fn(); torch.cuda.synchronize()
  g = torch.cuda.CUDAGraph()
  with torch.cuda.graph(g):
    for _ in range(8):
      fn()
  g.replay(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    g.replay()
  e1.record(); torch.cuda.synchronize()
  return 1e3 * e0.elapsed_time(e1) / (reps * 8)
P = C.c_void_p
for name, H, Ci, Co, pool in (('L4', 128, 16, 32, 1), ('L5', 128, 32, 32, 2), ('L6', 64, 32, 64, 2)):
  x = torch.randn(8, H, H, Ci, device='cuda')
  w = (np.random.RandomState(0).randn(3, 3, Ci, Co) / np.sqrt(9 * Ci)).astype(np.float32)
  sc, sh = [torch.from_numpy(a).cuda() for a in ops.fold_bn(None, Co)]
  ww = torch.from_numpy(ops.pack_wino_weights(w)).cuda()
  yo = torch.empty(8, H // pool, H // pool, Co, device='cuda')
  out = []
  for so in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wv', '*.so'))):
    lib = C.CDLL(so)
    f = lib.ra_conv_wino_f32
    f.restype = C.c_int
    f.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int, P, P, P, C.c_int, C.c_int, C.c_int, P, P]
    st = torch.cuda.current_stream
    call = lambda: f(x.data_ptr(), 8, H, H, Ci, ww.data_ptr(), sc.data_ptr(), sh.data_ptr(), Co, 1, pool, yo.data_ptr(),
                     torch.cuda.current_stream().cuda_stream)
    out.append('%s %.1f' % (os.path.basename(so)[5:-3], t_us(call)))
  print(name, ' | '.join(out))
