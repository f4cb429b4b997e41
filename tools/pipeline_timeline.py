#!/usr/bin/env python
"""Timeline statistics of a rocprofv3 kernel trace of the pipelined bench (4 batches in flight):
how much of the wall time has 0 / 1 / >=2 controller-CNN kernels running, and the same for any kernel.
usage: pipeline_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
from collections import Counter
# the pipeline's slots run on their own queues; the lone forward and the roofline replays use the null
# stream's queue (the busiest one).  Window = the second half of the slot queues' activity.
qcount = Counter(r['Queue_Id'] for r in rows)
main_q = qcount.most_common(1)[0][0]
slot = [r for r in rows if r['Queue_Id'] != main_q and 'ra::' in r['Kernel_Name']]
print('queues', dict(qcount), 'main', main_q)
lo = min(int(r['Start_Timestamp']) for r in slot); hi = max(int(r['End_Timestamp']) for r in slot)
t_lo = lo + (hi - lo) // 2
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in slot if int(r['Start_Timestamp']) >= t_lo]
ev.sort()
T = ev[-1][1] - ev[0][0]
is_enc = lambda n: ('conv_pair8_mfma' in n or 'conv_pair_persist' in n or 'conv3x3_mfma<16, 2' in n or 'conv3x3_mfma<16, 1, 4' in n)
def depth_hist(sel):
  pts = []
  for s, e, n in ev:
    if sel(n):
      pts.append((s, 1)); pts.append((e, -1))
  pts.sort()
  hist, d, last = {}, 0, ev[0][0]
  for t, k in pts:
    hist[d] = hist.get(d, 0) + (t - last)
    d += k; last = t
  hist[d] = hist.get(d, 0) + (ev[-1][1] - last)
  return {k: round(v / T, 3) for k, v in sorted(hist.items())}
print('window %.2f ms, %d kernels' % (T / 1e6, len(ev)))
print('controller-CNN kernels running concurrently (share of wall time):', depth_hist(is_enc))
print('any kernel:', depth_hist(lambda n: True))
enc_busy = sum(e - s for s, e, n in ev if is_enc(n))
print('sum of controller-CNN kernel durations / wall: %.3f' % (enc_busy / T))
oth = sum(e - s for s, e, n in ev if not is_enc(n))
print('sum of other kernel durations / wall: %.3f' % (oth / T))
