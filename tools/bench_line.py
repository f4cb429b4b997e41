#!/usr/bin/env python
"""bench.py's JSON line (a file argument, or stdin when there is none) -> a short summary (tuning aid)."""
import json, os, sys
src = open(sys.argv[1]).read() if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else sys.stdin.read()
d = json.loads(src.strip().splitlines()[-1])
c, r, a = d['config'], d.get('roofline', {}), d.get('roofline_attn', {})
print('in_flight', c.get('batches_in_flight'), 'value', round(d['value']), 'ms/step', round(d['ms_per_step'], 3),
      'lone', round(c.get('lone_batch_ms', 0), 3), 'enc_us', round(r.get('avg_us_per_launch_group', 0), 1), 'frac',
      round(r.get('frac', 0), 3), 'attn_us', round(a.get('avg_us_per_launch_group', 0), 2), 'frac', round(a.get('frac', 0), 3),
      'B32', round(a.get('at_B32', {}).get('frac_algorithmic', a.get('at_B32', {}).get('frac', 0)), 3), 'tail_us', round(d.get('tail_us', 0), 1))
