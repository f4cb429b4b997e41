#!/usr/bin/env python
"""Where do the __amd_rocclr_copyBuffer dispatches of a kernel trace sit?  Counts (previous kernel, next kernel)
pairs around every copy, per queue.  usage: copy_neighbours.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
short = lambda n: n.replace('void ', '').replace('ra::', '').split('(')[0][:48]
by = collections.Counter()
sizes = collections.Counter()
for i, r in enumerate(rows):
  if 'copyBuffer' in r['Kernel_Name']:
    prev = next((short(rows[j]['Kernel_Name']) for j in range(i - 1, -1, -1) if 'copyBuffer' not in rows[j]['Kernel_Name']), '-')
    nxt = next((short(rows[j]['Kernel_Name']) for j in range(i + 1, len(rows)) if 'copyBuffer' not in rows[j]['Kernel_Name']), '-')
    by[(prev, nxt)] += 1
    sizes[(r.get('Grid_Size', '?'), r.get('Workgroup_Size', '?'))] += 1
print('copies', sum(by.values()), 'of', len(rows), 'dispatches')
for (p, n), c in by.most_common(25):
  print('%6d  %-48s -> %s' % (c, p, n))
print('grid sizes:', sizes.most_common(8))
