#!/bin/bash
# Round 6: kernel-trace stats of bench.py --config cfg3|cfg5 (one slot in flight, so kernel durations are not inflated by overlap)
# and of the default pipelined run; the counters this box offers for MFMA instruction counts.
set -u
OUT=gpurun_out/${1:-r06a}
mkdir -p $OUT
export TMPDIR=/tmp
R="timeout 600 rocprofv3 --output-format csv"
for c in cfg3 cfg5; do
  $R --kernel-trace --stats -d $OUT/tr_$c -o t -- python bench.py --config $c --in-flight 1 --coalesce ${2:-0} --steps 4 --warmup 2 > $OUT/trace_$c.json 2> $OUT/trace_$c.log
  cp $(find $OUT/tr_$c -name '*kernel_stats.csv' | head -1) $OUT/r06_bench_${c}_kernel_stats.csv
  rm -rf $OUT/tr_$c
  timeout 600 python bench.py --config $c > $OUT/r06_bench_$c.json 2>> $OUT/bench.err
done
timeout 120 rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|MOPS" | head -60 > $OUT/counters_mfma.txt
timeout 300 python -m pytest tests/test_full_model_gpu.py -q -s -k "xcd_local" 2>&1 | tail -8 > $OUT/park_test.txt
ls -la $OUT
