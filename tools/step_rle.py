#!/usr/bin/env python
"""Run-length listing of the dispatches between the last two occurrences of a marker kernel in a kernel trace
(default marker: the augmentation's first launch = one training step), keeping only runs outside the big middle.
usage: step_rle.py <kernel_trace.csv> [marker substring] [head=60] [tail=60]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
mark = sys.argv[2] if len(sys.argv) > 2 else 'random_transform'
head = int(sys.argv[3]) if len(sys.argv) > 3 else 60
tail = int(sys.argv[4]) if len(sys.argv) > 4 else 60
idx = [i for i, r in enumerate(rows) if mark in r['Kernel_Name']]
# the marker comes in groups (4 gather launches): take the first of the last two groups
starts = [i for k, i in enumerate(idx) if k == 0 or idx[k - 1] < i - 8]
a, b = starts[-2], starts[-1]
short = lambda n: n.replace('void ', '').replace('ra::', '').split('(')[0][:60]
runs = []
for r in rows[a:b]:
  n = short(r['Kernel_Name'])
  if runs and runs[-1][0] == n:
    runs[-1][1] += 1
    runs[-1][2] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
  else:
    runs.append([n, 1, int(r['End_Timestamp']) - int(r['Start_Timestamp'])])
print('dispatches in the step:', b - a, ' copies:', sum(1 for r in rows[a:b] if 'copyBuffer' in r['Kernel_Name']),
      ' wall %.2f ms' % ((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e6))
for n, c, t in runs[:head]:
  print('%5d x %-60s %8.1f us' % (c, n, t / 1e3))
print('   ...')
for n, c, t in runs[-tail:]:
  print('%5d x %-60s %8.1f us' % (c, n, t / 1e3))
