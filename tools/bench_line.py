#!/usr/bin/env python
"""stdin: bench.py's JSON line -> a short summary (tuning aid)."""
import json, sys
d = json.loads(sys.stdin.read())
c, r = d['config'], d.get('roofline', {})
print(' '.join(sys.argv[1:]), 'in_flight', c.get('batches_in_flight'), 'value', round(d['value']), 'ms/step', round(d['ms_per_step'], 3),
      'lone', round(c.get('lone_batch_ms', 0), 3), 'enc_us', r.get('avg_us'), 'frac', r.get('frac'))
