import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('rec-attend-public_amd', 'oracle', 'tests', 'tools'):
  sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch
import full_model, ra_train, ra_ops as ops
import ra_oracle as ora, ra_oracle_torch as ort
import test_train_gpu as tt
cuda = torch.device('cuda')
opt, P, x, y_gt, s_gt = tt._case(T=3, wmul=0.6, fixed_order=False, **tt.KNOB_OPT)
B, T, H, W = 2, 3, 64, 64
rng = np.random.RandomState(5)
knobs = {'pad': rng.uniform(0.1, 0.3, (B, T, 1)), 'shift': rng.uniform(-0.05, 0.05, (B, T, 2)),
         'u_box': rng.rand(B, T, 1), 'u_segm': rng.rand(B, T, 1), 'segm_noise': 0.3 * rng.rand(T, B, H, W)}
K = ort.knob_setup(opt, y_gt.astype(np.float64), {k: np.asarray(v, np.float64) for k, v in knobs.items()}, 0)
m = full_model.get_model(opt).load_weights(P)
ts = ra_train.TrainStep(m)
kd = {k: torch.tensor(v, dtype=torch.float32, device=cuda) for k, v in knobs.items()}
yg = torch.tensor(y_gt, device=cuda)
ctr_n, size_n, kb, ks = ts._knob_setup(yg, kd)
print('ctr  prod', ctr_n.cpu().numpy().round(3).tolist()); print('ctr  orac', K['ctr'].numpy().round(3).tolist())
print('size prod', size_n.cpu().numpy().round(3).tolist()); print('size orac', K['size'].numpy().round(3).tolist())
print('kb', kb.flatten().tolist(), K['kb'].flatten().tolist(), 'ks', ks.flatten().tolist(), K['ks'].flatten().tolist())
_, box_gt = ops.gt_box(yg, float(opt['attn_box_padding_ratio']), float(opt['padding']) + 4.0)
print('box_gt diff', float(np.abs(box_gt.cpu().numpy() - K['box_gt']).max()), 'sums', box_gt.sum(dim=(2, 3)).cpu().numpy().tolist(), K['box_gt'].sum(axis=(2, 3)).tolist())
fwd, _ = ort.forward(opt, P, x, phase_train=True, knobs=knobs, y_gt=y_gt, global_step=0)
with torch.no_grad():
  loss, pieces, _ = ts.forward_loss(x, y_gt, s_gt, knobs=kd)
print('y_out diff per t', [float(np.abs(pieces['y_out'][:, t].cpu().numpy() - fwd['y_out'][:, t].numpy()).max()) for t in range(T)])
