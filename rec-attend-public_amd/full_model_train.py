#!/usr/bin/env python
"""Entry point with the flag surface of the reference's full_model_train.py (:460-668).

Builds the model from the same flags / `model_opt` and persists `model_opt.yaml` + the initial
weights under results/<model_id>/ (the layout full_model_eval.py restores from,
utils/saver.py:7-35 with .npz instead of a TF checkpoint).  The optimisation loop itself — losses,
Hungarian matching, Adam, the RCCL gradient all-reduce — is the training step, SURVEY.md §8(f)
rank 2, which is not built yet: without --init_only this exits with an explicit error."""
import argparse
import os
import sys

import numpy as np
import yaml

import cmd_args_parser as cap
import full_model


def build_parser():
  p = argparse.ArgumentParser(description='Train full model (recurrent attention)')
  for table in (cap.TRAIN_FLAGS, cap.DATA_FLAGS, cap.MODEL_FLAGS, cap.LEGACY_FLAGS):
    cap.add_flags(p, table)
  cap.add_size_overrides(p)
  p.add_argument('--init_only', action='store_true',
                 help='write model_opt.yaml + initial weights and stop')
  return p


def main(argv=None):
  args = build_parser().parse_args(argv)
  model_opt = cap.make_model_opt(args, args.inp_height, args.inp_width, args.timespan)
  model = full_model.get_model(model_opt, is_training=True)
  model_id = args.model_id or 'full_model'
  folder = os.path.join(args.results, model_id)
  os.makedirs(folder, exist_ok=True)
  with open(os.path.join(folder, 'model_opt.yaml'), 'w') as f:
    yaml.safe_dump(model_opt, f)
  np.savez(os.path.join(folder, 'weights.npz'), **model.state_dict_numpy())
  print('wrote %s (model_opt.yaml, weights.npz: %d tensors)' % (folder, len(model.weight_keys())))
  if not args.init_only:
    sys.exit('full_model_train: the training step (losses, matching, Adam, gradient all-reduce) '
             'is not built yet (SURVEY.md §8f rank 2); use --init_only to create a model folder '
             'for full_model_eval.py')


if __name__ == '__main__':
  main()
