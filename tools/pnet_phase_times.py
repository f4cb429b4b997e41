#!/usr/bin/env python
"""Per-phase durations of the K4 phase kernel from a rocprofv3 kernel trace of
`bench.py --fuse-patchnet` (RA_PNET_MODE=0: one launch per phase): the dispatches of
patchnet_kernel<false> repeat with the phase list's period."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'patchnet_kernel' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
nph = int(sys.argv[2]) if len(sys.argv) > 2 else 8
durs = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
durs = durs[len(durs) // 2 // nph * nph:]  # the steady-state half
per = [sum(durs[p::nph]) / max(1, len(durs[p::nph])) for p in range(nph)]
print('dispatches', len(durs), 'per-phase us', ['%.1f' % v for v in per], 'sum %.1f' % sum(per))
