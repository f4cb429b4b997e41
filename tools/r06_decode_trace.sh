#!/bin/bash
set -u
OUT=gpurun_out/${1:-r06b}
mkdir -p $OUT
export TMPDIR=/tmp
for c in cfg2 cfg3 cfg5; do
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$c -o t -- python tools/decode_trace.py run $c > $OUT/run_$c.log 2>&1
  f=$(find $OUT/tr_$c -name '*kernel_trace.csv' | head -1)
  python tools/decode_trace.py show $f > $OUT/r06_decode_sequence_$c.txt 2>&1
  grep -c copyBuffer $f >> $OUT/r06_decode_sequence_$c.txt
  rm -rf $OUT/tr_$c
done
