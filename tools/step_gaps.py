#!/usr/bin/env python
"""Idle time of the GPU inside and between training steps, from a rocprofv3 kernel trace of `bench.py --train`:
the steady-state steps are found by the Hungarian launch (one per step); prints per step its wall time (Hungarian start to
Hungarian start), the summed kernel durations, and the largest gaps between consecutive launches with the kernels around
them.  usage: step_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
  for r in csv.DictReader(f):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'hungarian_kernel' in r[2]]
marks = marks[-5:]  # the last steps are replays of the captured graph
for a, b in zip(marks[:-1], marks[1:]):
  seg = rows[a:b + 1]
  wall = (seg[-1][0] - seg[0][0]) / 1e3
  busy = sum(e - s for s, e, _ in seg[:-1]) / 1e3
  gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:50], seg[i + 1][2][:50]) for i in range(len(seg) - 1))
  big = [g for g in gaps if g[0] > 15.0]
  print('step: wall %.0f us, kernel time %.0f us, idle %.0f us; gaps > 15 us: %d (%.0f us)' % (wall, busy, wall - busy, len(big), sum(g[0] for g in big)))
  for g in gaps[-6:]:
    print('   %7.1f us  after %-50s before %s' % g)
