#!/bin/bash
# usage: tools/r06_decode_trace.sh OUTDIR "cfg[:images] ..."   — launch sequence of one lone forward per configuration
set -u
OUT=gpurun_out/${1:-r06b}
mkdir -p $OUT
export TMPDIR=/tmp
for ci in ${2:-cfg2 cfg3 cfg5}; do
  c=${ci%%:*}; n=""; [ "$ci" != "$c" ] && n=${ci#*:}
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/tr_$c -o t -- python tools/decode_trace.py run $c $n > $OUT/run_$c.log 2>&1
  f=$(find $OUT/tr_$c -name '*kernel_trace.csv' | head -1)
  python tools/decode_trace.py show $f > $OUT/r06_decode_sequence_${c}_${n:-B}.txt 2>&1
  rm -rf $OUT/tr_$c
done
