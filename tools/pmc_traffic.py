#!/usr/bin/env python
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --pmc-group REPS` into
profiles/r01_pmc_encoder_traffic.json.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <REPS>
       <images> <size> <out.json> [kernel-substring,kernel-substring,...]
The optional last argument selects the kernels of the group (default: the controller-CNN kernels
ra::conv:: / ra::cpair:: / ra::wino::; e.g. "ra::attnd::" for the extract + paste pair).

FETCH_SIZE / WRITE_SIZE are in KiB per dispatch.  Per MI355X_MICROARCH.md (HBM section) gfx950's
FETCH_SIZE tallies 128-B requests as 64 B, so it is doubled; WRITE_SIZE is taken as is.  Only the
controller-CNN kernels (ra::conv::*, ra::cpair::*) launched AFTER the warm-up forward are summed:
the forward pass launches each of them a known number of times, which is subtracted by taking the
last REPS x launches_per_group dispatches of the run.
"""
import csv, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_stamp():
  """What the counters were collected with: bench.py reports whether the file it replays still matches the build."""
  sha = lambda p: hashlib.sha256(open(p, 'rb').read()).hexdigest()[:16] if os.path.exists(p) else None
  return {'bench_py_sha16': sha(os.path.join(ROOT, 'bench.py')),
          'librecattend_sha16': sha(os.path.join(ROOT, 'rec-attend-public_amd', 'librecattend.so'))}


KEYS = ['ra::conv::', 'ra::cpair::', 'ra::wino::', 'ra::csplit::']


def tail_sum(path, counter, reps):
  rows = [r for r in csv.DictReader(open(path))
          if r['Counter_Name'] == counter and any(k in r['Kernel_Name'] for k in KEYS)]
  rows.sort(key=lambda r: int(r['Dispatch_Id']))
  # the eager group launches are the tail of the run; find the group length from the repeating
  # kernel-name pattern at the end
  names = [r['Kernel_Name'] for r in rows]
  for glen in range(1, 64):
    if names[-glen:] == names[-2 * glen:-glen] and len(set(names[-glen:])) > 1:
      break
  else:
    raise SystemExit('no repeating launch group found')
  tail = rows[-glen * reps:]
  per_kernel = {}
  for r in tail:
    per_kernel.setdefault(r['Kernel_Name'][:80], []).append(float(r['Counter_Value']))
  return glen, sum(float(r['Counter_Value']) for r in tail) / reps, \
      {k: sum(v) / len(v) for k, v in per_kernel.items()}


def main():
  fcsv, wcsv, reps, images, size, out = sys.argv[1:7]
  if len(sys.argv) > 7:
    KEYS[:] = sys.argv[7].split(',')
  reps = int(reps)
  glen, fetch_kib, fk = tail_sum(fcsv, 'FETCH_SIZE', reps)
  glen2, write_kib, wk = tail_sum(wcsv, 'WRITE_SIZE', reps)
  assert glen == glen2
  rec = {
      'images': int(images), 'size': int(size), 'launches_per_group': glen, 'reps': reps,
      'fetch_kib_raw_per_group': fetch_kib, 'write_kib_raw_per_group': write_kib,
      'fetch_bytes_per_group': fetch_kib * 1024 * 2, 'write_bytes_per_group': write_kib * 1024,
      'hbm_bytes_per_launch_group': fetch_kib * 1024 * 2 + write_kib * 1024,
      'correction': 'FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B), WRITE_SIZE x1',
      'per_kernel_mean_kib': {'FETCH_SIZE': fk, 'WRITE_SIZE': wk},
      'collected_with': build_stamp(),
  }
  json.dump(rec, open(out, 'w'), indent=1)
  print(json.dumps(rec)[:600])


if __name__ == '__main__':
  main()
