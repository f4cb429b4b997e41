#!/usr/bin/env python
"""Group a rocprofv3 kernel_stats.csv of the training bench by kernel family.  argv: csv, steps traced."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
fam = {}
for r in rows:
  n = r['Name']
  if 'ra::train::wgrad' in n: k = 'wgrad'
  elif 'hungarian' in n: k = 'hungarian'
  elif 'ra::train::bn' in n or 'chan_' in n: k = 'bn kernels'
  elif 'ra::conv' in n or 'ra::cpair' in n: k = 'conv fwd/dgrad'
  elif 'ra::train' in n: k = 'train other'
  elif n.startswith('Cijk'): k = 'rocBLAS gemm'
  elif 'at::native' in n: k = 'torch elementwise/reduce'
  elif 'copyBuffer' in n or 'fillBuffer' in n: k = 'copies'
  elif 'ra::' in n: k = 'ra other'
  else: k = 'other'
  f = fam.setdefault(k, [0, 0]); f[0] += int(r['TotalDurationNs']); f[1] += int(r['Calls'])
tot = sum(v[0] for v in fam.values()); calls = sum(v[1] for v in fam.values())
print('total %.1f ms/step, %d launches/step' % (tot / steps / 1e6, calls / steps))
for k, (t, c) in sorted(fam.items(), key=lambda x: -x[1][0]):
  print('%-28s %8.2f ms/step %7d calls/step' % (k, t / steps / 1e6, c / steps))
if len(sys.argv) > 3:
  for r in sorted(rows, key=lambda r: -int(r['TotalDurationNs']))[:int(sys.argv[3])]:
    print('%7d %8.2f ms/step %7.1f us  %s' % (int(r['Calls']) / steps, int(r['TotalDurationNs']) / steps / 1e6, float(r['AverageNs']) / 1e3, r['Name'].replace('at::native::', '')[:110]))
