#!/usr/bin/env python
"""Probe: when does the HIP runtime read GPU_MAX_HW_QUEUES — at library load (import torch) or at
the first HIP call?  argv[1] = 'before' | 'after' (set the variable before / after import torch)."""
import os, sys
when, q = sys.argv[1], sys.argv[2]
if when == 'before':
  os.environ['GPU_MAX_HW_QUEUES'] = q
import torch
if when == 'after':
  os.environ['GPU_MAX_HW_QUEUES'] = q
sys.argv = [sys.argv[0]] + sys.argv[3:]
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import runpy
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'bench.py'), run_name='__main__')
