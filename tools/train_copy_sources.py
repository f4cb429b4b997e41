#!/usr/bin/env python
"""Which host ops issue the training step's device-to-device copies?  One EAGER step under torch.profiler (all
threads, so the autograd engine's worker is included): every aten::copy_ / aten::clone with the chain of profiler
ranges it ran under."""
import collections, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'rec-attend-public_amd'))
import numpy as np
import torch
import bench, full_model, full_model_train as fmt, ra_train

B, T, S = 8, 16, 512
opt = bench.make_opt('cvppp', S, S, T)
opt.update(use_knob=True, knob_base=1.0, knob_decay=0.9, steps_per_knob_decay=300, knob_box_offset=300,
           knob_segm_offset=500, knob_use_timescale=True, gt_box_ctr_noise=0.05, gt_box_pad_noise=0.1,
           gt_segm_noise=0.3, base_learn_rate=1e-3, learn_rate_decay=0.96, steps_per_learn_rate_decay=5000)
ra_train.TrainStep.use_graph = False
model = full_model.get_model(opt, is_training=True)
rng = np.random.RandomState(1234)
x, y_gt, s_gt = fmt.synthetic_batch(rng, B, S, S, T)
gen = torch.Generator(device='cuda').manual_seed(1234)
feed = {'x': torch.as_tensor(x).cuda(), 'y_gt': torch.as_tensor(y_gt).cuda(), 's_gt': torch.as_tensor(s_gt).cuda(),
        'phase_train': True, 'generator': gen}
for _ in range(2):
  model.run(['loss', 'train_step'], feed)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
  model.run(['loss', 'train_step'], feed)
  torch.cuda.synchronize()
by = collections.Counter()
kern = collections.Counter()
for e in prof.events():
  if e.device_type == torch.autograd.DeviceType.CUDA or str(e.device_type).endswith('CUDA'):
    kern[e.name[:60]] += 1
    continue
  if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::_to_copy'):
    chain, p = [], e.cpu_parent
    while p is not None and len(chain) < 6:
      chain.append(p.name[:50])
      p = p.cpu_parent
    by[(e.name, ' < '.join(chain))] += 1
for (name, chain), n in by.most_common(40):
  print('%5d  %-16s %s' % (n, name, chain))
print('--- device-side events with "opy" in the name:')
for k, n in kern.most_common():
  if 'opy' in k or 'emcpy' in k:
    print('%5d  %s' % (n, k))
