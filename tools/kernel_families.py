#!/usr/bin/env python
"""Group the training bench's kernels by family.
  kernel_families.py <kernel_trace.csv> [top N kernels]   ONE steady-state step: the dispatches between the last two
                                                          augmentation launches of the trace (what a graph replay +
                                                          the optimizer step issue; model set-up and the eager first
                                                          step are outside)
  kernel_families.py <kernel_stats.csv> <steps> [top N]   the whole run divided by `steps` (older summaries)"""
import collections, csv, sys


def family(n):
  if 'ra::train::wgrad' in n: return 'wgrad'
  if 'hungarian' in n: return 'hungarian'
  if 'ra::train::bn' in n or 'chan_' in n or 'moments_from_partials' in n: return 'bn kernels'
  if 'ra::conv' in n or 'ra::cpair' in n or 'ra::csplit' in n: return 'conv fwd/dgrad'
  if 'ra::train' in n: return 'train other'
  if n.startswith('Cijk') or 'rocblas' in n: return 'rocBLAS gemm / gemv'
  if 'at::native' in n: return 'torch elementwise/reduce'
  if 'copyBuffer' in n or 'fillBuffer' in n: return 'copies / fills'
  if 'ra::' in n: return 'ra other'
  return 'other'


rows = list(csv.DictReader(open(sys.argv[1])))
if 'Kernel_Name' in rows[0]:
  rows.sort(key=lambda r: int(r['Start_Timestamp']))
  idx = [i for i, r in enumerate(rows) if 'random_transform' in r['Kernel_Name']]
  starts = [i for k, i in enumerate(idx) if k == 0 or idx[k - 1] < i - 8]
  a, b = starts[-2], starts[-1]
  fam, per = {}, collections.OrderedDict()
  for r in rows[a:b]:
    n, d = r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    f = fam.setdefault(family(n), [0, 0]); f[0] += d; f[1] += 1
    p = per.setdefault(n, [0, 0]); p[0] += d; p[1] += 1
  wall = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6
  tot = sum(v[0] for v in fam.values())
  print('one steady-state step: %d dispatches, %.1f ms of kernel time, %.1f ms wall under the tracer' % (b - a, tot / 1e6, wall))
  for k, (t, c) in sorted(fam.items(), key=lambda x: -x[1][0]):
    print('%-28s %8.2f ms/step %7d calls/step' % (k, t / 1e6, c))
  if len(sys.argv) > 2:
    for n, (t, c) in sorted(per.items(), key=lambda x: -x[1][0])[:int(sys.argv[2])]:
      print('%7d %8.2f ms/step %7.1f us  %s' % (c, t / 1e6, t / c / 1e3, n.replace('at::native::', '').replace('void ', '')[:110]))
  sys.exit(0)
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
fam = {}
for r in rows:
  f = fam.setdefault(family(r['Name']), [0, 0]); f[0] += int(r['TotalDurationNs']); f[1] += int(r['Calls'])
tot = sum(v[0] for v in fam.values()); calls = sum(v[1] for v in fam.values())
print('total %.1f ms/step, %d launches/step' % (tot / steps / 1e6, calls / steps))
for k, (t, c) in sorted(fam.items(), key=lambda x: -x[1][0]):
  print('%-28s %8.2f ms/step %7d calls/step' % (k, t / steps / 1e6, c / steps))
if len(sys.argv) > 3:
  for r in sorted(rows, key=lambda r: -int(r['TotalDurationNs']))[:int(sys.argv[3])]:
    print('%7d %8.2f ms/step %7.1f us  %s' % (int(r['Calls']) / steps, int(r['TotalDurationNs']) / steps / 1e6, float(r['AverageNs']) / 1e3, r['Name'].replace('at::native::', '')[:110]))
