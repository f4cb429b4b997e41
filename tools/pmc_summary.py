#!/usr/bin/env python
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.Counter())
for r in rows:
  k = r['Kernel_Name']
  agg[k][r['Counter_Name']] += float(r['Counter_Value'])
  cnt[k][r['Counter_Name']] += 1
names = sorted({c for v in agg.values() for c in v})
print('kernel,dispatches,' + ','.join(names))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
  if not k.startswith(('void ra::', 'ra::')):
    continue
  n = max(cnt[k].values())
  print('"%s",%d,' % (k[:110], n) + ','.join('%.0f' % (v.get(c, 0.0) / max(cnt[k][c], 1)) for c in names))
