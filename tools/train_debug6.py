"""Debug aid (GPU): the controller-CNN chain alone, HIP function vs torch stand-in, same input and
upstream gradient; per-layer activations and gradients."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('rec-attend-public_amd', 'oracle', 'tests', 'tools'):
  sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import torch.nn.functional as F
import full_model, ra_train
import test_train_gpu as tt

def conv_ref(x, w, b, gamma, beta, meta):
  pool = meta['pool']
  u = F.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1) + b
  mean = u.mean(dim=(0, 1, 2)); var = ((u - mean) ** 2).mean(dim=(0, 1, 2))
  v = torch.relu((u - mean) * torch.rsqrt(var + 1e-3) * gamma + beta)
  y = F.max_pool2d(v.permute(0, 3, 1, 2), pool, pool).permute(0, 2, 3, 1) if pool == 2 else v
  return y, mean.detach(), var.detach()

opt, P, x, y_gt, s_gt = tt._case(T=2)
dev = torch.device('cuda')
pools = opt['ctrl_cnn_pool']
rng = np.random.RandomState(5)
x0 = torch.tensor(np.concatenate([x, np.zeros(x.shape[:3] + (1,), np.float32)], axis=3), device=dev)
res = {}
for mode in ('hip', 'ref'):
  leaves = {k: torch.tensor(P[k], device=dev, requires_grad=True) for k in P if k.startswith('ctrl_cnn') and 'ema' not in k}
  h, acts = x0, []
  for i in range(8):
    meta = dict(transposed=False, stride=1, pool=pools[i], relu=True, chan_map=None)
    args = (h, leaves['ctrl_cnn_w_%d' % i], leaves['ctrl_cnn_b_%d' % i], leaves['ctrl_cnn_%d_0_gamma' % i], leaves['ctrl_cnn_%d_0_beta' % i], meta)
    h = (ra_train.ConvBNActPool.apply(*args) if mode == 'hip' else conv_ref(*args))[0]
    h.retain_grad()
    acts.append(h)
  g = torch.tensor(np.random.RandomState(9).randn(*h.shape).astype(np.float32), device=dev)
  (h * g).sum().backward()
  res[mode] = (acts, leaves)
rel = lambda a, b: float((a - b).abs().max() / max(float(b.abs().max()), 1e-9))
for i in range(8):
  ah, ar = res['hip'][0][i], res['ref'][0][i]
  lh, lr = res['hip'][1], res['ref'][1]
  print('L%d act %.1e  dact %.1e  dw %.1e  dgamma %.1e  dbeta %.1e   (zeros in act: %.2f)' % (
      i, rel(ah, ar), rel(ah.grad, ar.grad), rel(lh['ctrl_cnn_w_%d' % i].grad, lr['ctrl_cnn_w_%d' % i].grad),
      rel(lh['ctrl_cnn_%d_0_gamma' % i].grad, lr['ctrl_cnn_%d_0_gamma' % i].grad),
      rel(lh['ctrl_cnn_%d_0_beta' % i].grad, lr['ctrl_cnn_%d_0_beta' % i].grad), float((ar == 0).float().mean())))
