#!/usr/bin/env python
"""Which host-side ops launch the training step's small kernels?  One EAGER step (TrainStep.use_graph off)
under torch.profiler with Python stacks; prints aten ops by call count with the ra_train.py line that issued them."""
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
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

by = collections.Counter()


class Census(TorchDispatchMode):
  def __torch_dispatch__(self, func, types, args=(), kwargs=None):
    site = 'autograd engine (no Python frame)'
    for fr in reversed(traceback.extract_stack(limit=40)):
      fn = os.path.basename(fr.filename)
      if fn in ('ra_train.py', 'ra_ops.py', 'modellib.py', 'nnlib.py', 'full_model.py'):
        site = '%s:%d %s' % (fn, fr.lineno, (fr.line or '').strip()[:80])
        break
    by[(str(func).replace('aten.', ''), site)] += 1
    return func(*args, **(kwargs or {}))


with Census():
  model.run(['loss', 'train_step'], feed)
torch.cuda.synchronize()
skip = ('view', 'reshape', 'detach', 'slice', 'select', 'expand', 'permute', 'transpose', 't.default', 'unsqueeze', 'squeeze',
        'alias', 'as_strided', '_unsafe_view', 'unbind', 'split', 'empty', 'is_', 'stride', 'size', 'sym_', 'numel', 'dim')
rows = [(n, k) for k, n in by.items() if not any(k[0].startswith(x) for x in skip)]
rows.sort(reverse=True)
print('aten calls of one eager training step (views and allocations dropped), by (op, issuing line):')
for n, (name, site) in rows[:90]:
  print('%6d  %-30s %s' % (n, name, site))
print('total', sum(n for n, _ in rows))
