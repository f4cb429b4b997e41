#!/usr/bin/env python
"""Print a window of a rocprofv3 kernel trace (CSV) as a per-kernel timeline.
usage: timeline.py <kernel_trace.csv> [skip_from_end=600] [count=120]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 600
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 120
win = rows[len(rows) - skip:len(rows) - skip + cnt]
t0 = int(win[0]['Start_Timestamp'])
def short(n):
  n = n.replace('void ', '').replace('ra::', '')
  return n.split('(')[0][:44]
for r in win:
  s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
  print('%-44s q%-2s %9.1f %9.1f %7.1f' % (short(r['Kernel_Name']), r['Queue_Id'], s / 1e3, e / 1e3, (e - s) / 1e3))
