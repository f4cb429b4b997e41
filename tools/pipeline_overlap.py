#!/usr/bin/env python
"""What the decode pipeline's kernels do to each other — from a rocprofv3 kernel trace of a pipelined bench.py run.
usage: pipeline_overlap.py <kernel_trace.csv> [steady_fraction=0.5]

Takes the middle `steady_fraction` of the trace's wall time (the pipelined region: the lone forwards and probes of bench.py run
before and after it, so pass --steps large enough that the middle is pipeline) and prints
  * the share of wall time with an encoder (controller-CNN) kernel running, with only tail kernels running, with nothing running;
  * per encoder kernel: its duration when no controller kernel overlaps it against when one does;
  * per tail kernel: the gap between the end of its stream predecessor and its own start (queueing behind other streams' work).
"""
import csv, sys, collections

ENC = ('conv_pair8_mfma', 'conv_pair_wino_mfma', 'conv_wino_mfma', 'conv_split_kernel', 'conv3x3_mfma<16, 1, 4, 2, 1, false')
CTRL = ('controller_batch_kernel', 'controller_split_kernel', 'controller_kernel')


def short(n):
  return n.replace('void ', '').replace('ra::', '').split('(')[0][:52]


def union(iv):
  iv = sorted(iv)
  out = []
  for s, e in iv:
    if out and s <= out[-1][1]:
      out[-1][1] = max(out[-1][1], e)
    else:
      out.append([s, e])
  return out


def length(iv):
  return sum(e - s for s, e in iv)


def overlap(s, e, iv):  # ns of [s, e) covered by the sorted disjoint intervals iv
  t = 0
  for a, b in iv:
    if b <= s:
      continue
    if a >= e:
      break
    t += min(b, e) - max(a, s)
  return t


def main():
  rows = list(csv.DictReader(open(sys.argv[1])))
  frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
  for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  rows.sort(key=lambda r: r['s'])
  t0, t1 = rows[0]['s'], rows[-1]['e']
  mid = (t0 + t1) / 2
  lo, hi = mid - frac * (t1 - t0) / 2, mid + frac * (t1 - t0) / 2
  win = [r for r in rows if r['s'] >= lo and r['e'] <= hi]
  queues = sorted(set(r['Queue_Id'] for r in win))
  print('window %.2f ms, %d dispatches on queues %s' % ((hi - lo) / 1e6, len(win), queues))
  is_enc = lambda r: any(k in r['Kernel_Name'] for k in ENC)
  is_ctrl = lambda r: any(k in r['Kernel_Name'] for k in CTRL)
  enc = union([(r['s'], r['e']) for r in win if is_enc(r)])
  ctrl = union([(r['s'], r['e']) for r in win if is_ctrl(r)])
  allk = union([(r['s'], r['e']) for r in win])
  W = hi - lo
  print('encoder kernel running   %.3f of wall' % (length(enc) / W))
  print('any kernel running       %.3f of wall   (tail-only: %.3f, idle: %.3f)' %
        (length(allk) / W, (length(allk) - length(enc)) / W, 1 - length(allk) / W))
  print('controller running       %.3f of wall' % (length(ctrl) / W))
  # how many encoder kernels run at the same time
  ev = []
  for r in win:
    if is_enc(r):
      ev += [(r['s'], 1), (r['e'], -1)]
  ev.sort()
  depth, last, hist = 0, lo, collections.Counter()
  for t, d in ev:
    hist[depth] += t - last
    last, depth = t, depth + d
  print('encoder kernels in flight (share of wall): ' + '  '.join('%d: %.3f' % (k, v / W) for k, v in sorted(hist.items())))
  print()
  print('%-52s %6s %9s %9s | %6s %9s' % ('encoder kernel', 'n', 'alone us', 'sum us', 'n', 'w/ ctrl'))
  by = collections.defaultdict(lambda: [[], []])
  for r in win:
    if is_enc(r):
      ov = overlap(r['s'], r['e'], ctrl)
      by[short(r['Kernel_Name'])][1 if ov > 0.3 * (r['e'] - r['s']) else 0].append((r['e'] - r['s']) / 1e3)
  med = lambda v: sorted(v)[len(v) // 2] if v else float('nan')
  for k, (a, b) in sorted(by.items()):
    print('%-52s %6d %9.1f %9.0f | %6d %9.1f' % (k, len(a), med(a), sum(a) + sum(b), len(b), med(b)))
  print()
  # per queue: gaps between consecutive kernels of the same queue (the wait for a place on the chip + launch latency)
  print('%-52s %6s %9s %9s %9s' % ('kernel (same-queue gap before it)', 'n', 'med gap', 'med dur', 'sum us'))
  gaps = collections.defaultdict(lambda: [[], []])
  prev = {}
  for r in win:
    q = r['Queue_Id']
    if q in prev:
      gaps[short(r['Kernel_Name'])][0].append((r['s'] - prev[q]) / 1e3)
    gaps[short(r['Kernel_Name'])][1].append((r['e'] - r['s']) / 1e3)
    prev[q] = r['e']
  tot = 0
  for k, (g, d) in sorted(gaps.items(), key=lambda kv: -sum(kv[1][0]) - sum(kv[1][1])):
    print('%-52s %6d %9.2f %9.1f %9.0f' % (k, len(d), med(g), med(d), sum(g) + sum(d)))
  print()
  print('per queue: busy share (its kernels + gaps < 20 us) is ~1 for a stream that never waits for the host')
  for q in queues:
    rs = [r for r in win if r['Queue_Id'] == q]
    print('  queue %s: %d dispatches, kernel time %.3f of wall' % (q, len(rs), sum(r['e'] - r['s'] for r in rs) / W))


if __name__ == '__main__':
  main()
