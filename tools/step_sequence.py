#!/usr/bin/env python
"""The dispatches of ONE steady-state training step in launch order (short names, durations), from a rocprofv3
kernel_trace.csv of `bench.py --train`: what sits between the ra:: kernels (ATen element-wise chains, copies, library
GEMMs).  step_sequence.py <kernel_trace.csv> [only-non-ra]"""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'random_transform' in r['Kernel_Name']]
starts = [i for k, i in enumerate(idx) if k == 0 or idx[k - 1] < i - 8]
a, b = starts[-2], starts[-1]


def short(n):
  n = n.replace('void ', '').replace('at::native::', '')
  m = re.search(r'ra::\w+::\w+(<[^>]*>)?', n)
  if m:
    return m.group(0)
  m = re.search(r'(\w+Functor|\w+_kernel_cuda|\w+Op|CatArrayBatchedCopy\w*|reduce_kernel|Cijk\w{0,12}|copyBuffer|fillBuffer|rocblas\w+)', n)
  tag = m.group(0) if m else n[:60]
  m2 = re.search(r'(vectorized|unrolled|manual_unroll|index_elementwise|reduce_kernel)', n)
  return (m2.group(0) + ':' if m2 and m2.group(0) not in tag else '') + tag


only = len(sys.argv) > 2
prev, cnt, dur = None, 0, 0
out = []
for r in rows[a:b]:
  s = short(r['Kernel_Name'])
  d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
  if s == prev:
    cnt += 1; dur += d
  else:
    if prev is not None:
      out.append((prev, cnt, dur))
    prev, cnt, dur = s, 1, d
out.append((prev, cnt, dur))
for s, c, d in out:
  if only and s.startswith('ra::'):
    print('  ..', s[:50]) if False else None
    continue
  print('%4d x %8.1f us  %s' % (c, d, s))
