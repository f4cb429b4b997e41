"""Debug aid (GPU): the same training graph with (A) the HIP functions and (B) plain torch float32
stand-ins for them, both against the float64 oracle: separates kernel bugs from float32 conditioning."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('rec-attend-public_amd', 'oracle', 'tests'):
  sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import torch.nn.functional as F
import full_model, ra_train
import ra_oracle_torch as ort
import test_train_gpu as tt

def conv_ref(x, w, b, gamma, beta, meta):
  tr, stride, pool = meta['transposed'], meta['stride'], meta['pool']
  cin_w = w.shape[3] if tr else w.shape[2]
  x = x[..., :cin_w]
  xi = x.permute(0, 3, 1, 2)
  H, W = x.shape[1], x.shape[2]
  if tr:
    wt = w.permute(3, 2, 0, 1)
    u = F.conv_transpose2d(xi, wt, stride=1, padding=1) if stride == 1 else F.conv_transpose2d(xi, wt, stride=2, padding=0)[:, :, :2*H, :2*W]
  else:
    u = F.conv2d(xi, w.permute(3, 2, 0, 1), padding=1)
  u = u.permute(0, 2, 3, 1) + b
  mean = u.mean(dim=(0, 1, 2)); var = ((u - mean) ** 2).mean(dim=(0, 1, 2))
  v = torch.relu((u - mean) * torch.rsqrt(var + 1e-3) * gamma + beta)
  y = F.max_pool2d(v.permute(0, 3, 1, 2), pool, pool).permute(0, 2, 3, 1) if pool == 2 else v
  return y, mean.detach(), var.detach()

class FakeApply:
  def __init__(self, fn): self.apply = fn

opt, P, x, y_gt, s_gt = tt._case(T=int(os.environ.get('T', '2')), H=int(os.environ.get('S', '64')), W=int(os.environ.get('S', '64')), wmul=float(os.environ.get('WMUL', '1')))
head, gref, stats = tt._oracle_grads(opt, P, x, y_gt, s_gt)
wd = float(opt['weight_decay'])
res = {}
HIP = (ra_train.ConvBNActPool, ra_train.PairIoU)
for mode in ('hip', 'torch_conv'):
  ra_train.ConvBNActPool = HIP[0] if mode in ('hip', 'torch_iou') else FakeApply(conv_ref)
  ra_train.PairIoU = HIP[1] if mode in ('hip', 'torch_conv') else FakeApply(ort.iou_pairwise)
  m = full_model.get_model(opt).load_weights(P)
  ts = ra_train.TrainStep(m)
  ts.bucket.zero_grad()
  loss, pieces, st = ts.forward_loss(x, y_gt, s_gt)
  loss.backward()
  res[mode] = {k: ts.bucket.grad_of[k].cpu().numpy() + (wd * P[k] if ra_train.is_decayed(k) else 0) for k in gref}
  errs = sorted(((float(np.abs(res[mode][k] - gref[k]).max() / max(np.abs(gref[k]).max(), 1e-3)), k) for k in gref), reverse=True)
  print(mode, 'loss %.6f' % float(loss), 'vs oracle:', [(k, '%.3f' % e) for e, k in errs[:6]])
for a, b in (('hip', 'torch_conv'),):
  errs = sorted(((float(np.abs(res[a][k] - res[b][k]).max() / max(np.abs(gref[k]).max(), 1e-3)), k) for k in gref), reverse=True)
  print(a, 'vs', b, [(k, '%.3f' % e) for e, k in errs[:6]])
