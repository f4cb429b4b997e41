"""Debug aid (GPU): product vs oracle training step, loss pieces and largest parameter / gradient
differences.  Test infrastructure only (imports oracle/)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('rec-attend-public_amd', 'oracle', 'tests'):
  sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import full_model, ra_train
import test_train_gpu as tt

opt, P, x, y_gt, s_gt = tt._case(T=int(os.environ.get('T', '2')))
m = full_model.get_model(opt).load_weights(P)
feed = {'x': x, 'y_gt': y_gt, 's_gt': s_gt, 'phase_train': True}
Pr = {k: v.astype(np.float64) for k, v in P.items()}
mom = {k: (np.zeros_like(v), np.zeros_like(v)) for k, v in Pr.items()}
for t in range(1, 4):
  head, gref, stats = tt._oracle_grads(opt, {k: v.astype(np.float32) for k, v in Pr.items()}, x, y_gt, s_gt)
  ts = m.trainer if getattr(m, 'trainer', None) else None
  names = ['loss', 'iou_soft', 'iou_soft_box', 'conf_loss', 'train_step']
  out = dict(zip(names, m.run(names, feed)))
  print('step', t, 'product', {k: round(float(out[k]), 5) for k in names[:-1]}, 'oracle',
        {k: round(float(head[k]), 5) for k in names[:-1]})
  g = {k: m.trainer.bucket.grad_of[k].cpu().numpy() for k in gref}
  wd = float(opt['weight_decay'])
  rows = []
  for k in gref:
    got = g[k] + (wd * Pr[k] if ra_train.is_decayed(k) else 0)
    rows.append((float(np.abs(got - gref[k]).max()), float(np.abs(gref[k]).max()), k))
  rows.sort(reverse=True)
  print('  worst grads (abs err, ref scale):', [(k, '%.2e' % e, '%.2e' % s) for e, s, k in rows[:8]])
  lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
  for k, gg in gref.items():
    gg = np.clip(gg, -1, 1)
    m1, v1 = mom[k]
    m1 = 0.9 * m1 + 0.1 * gg
    v1 = 0.999 * v1 + 0.001 * gg * gg
    mom[k] = (m1, v1)
    Pr[k] = Pr[k] - lr_t * m1 / (np.sqrt(v1) + 1e-7)
  got = m.state_dict_numpy()
  rows = sorted(((float(np.abs(got[k] - Pr[k]).max()), k) for k in gref), reverse=True)
  print('  worst params after step:', [(k, '%.2e' % e) for e, k in rows[:6]])
