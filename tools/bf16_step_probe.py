#!/usr/bin/env python
"""How far is the bf16 training step from its oracle (the float64 graph with bf16-rounded conv operands) and from the
unrounded float64 oracle, next to the float32 step — for several depths T, weight gains and seeds of the randomly
initialised test network (tests/test_train_gpu.py::_case).  Gradient cosines, losses, worst BatchNorm statistic."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ('tests', 'oracle', 'rec-attend-public_amd', ''):
  sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np, torch
import test_train_gpu as T
import ra_oracle_torch as ort, ra_train, full_model
for (tt, wm, seed) in [(3, 0.6, 3), (2, 0.6, 3), (3, 0.3, 3), (2, 0.3, 3), (3, 0.15, 3), (3, 0.3, 5), (2, 0.3, 5)]:
  opt, P, x, y_gt, s_gt = T._case(wmul=wm, T=tt, seed=seed)
  head64, gref64, _ = T._oracle_grads(opt, P, x, y_gt, s_gt)
  ort.set_conv_operands('bf16')
  head, gref, stats = T._oracle_grads(opt, P, x, y_gt, s_gt)
  ort.set_conv_operands(None)
  res = []
  for cd in ('bf16', 'float32'):
    m = full_model.get_model(dict(opt, compute_dtype=cd)).load_weights(P)
    ts = ra_train.TrainStep(m)
    ts.bucket.zero_grad()
    loss, pieces, st = ts.forward_loss(x, y_gt, s_gt)
    loss.backward()
    wd = float(opt['weight_decay'])
    got_of = lambda k: ts.bucket.grad_of[k].cpu().numpy()
    worst = max(max(T._rel(st[k][0].cpu().numpy(), mv[0].numpy()), T._rel(st[k][1].cpu().numpy(), mv[1].numpy())) for k, mv in stats.items())
    res.append('%s: cos(emul) %.4f cos(f64) %.4f loss %.5f worst stat %.3g' % (cd, T._grad_cosine(gref, got_of, P, wd), T._grad_cosine(gref64, got_of, P, wd), float(loss.detach()), worst))
  print('T=%d wmul=%.2f seed=%d  oracle loss emul %.5f f64 %.5f | %s | %s' % (tt, wm, seed, float(head['loss']), float(head64['loss']), res[0], res[1]), flush=True)
