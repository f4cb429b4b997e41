#!/bin/bash
set -u
OUT=gpurun_out/${1:-r06c}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv_split" 2>&1 | tail -8 > $OUT/k1s_tests.txt
timeout 900 python -m pytest tests/test_full_model_gpu.py -x -q -k "kitti or cityscapes or cfg3 or cfg5 or cfg1" 2>&1 | tail -8 >> $OUT/k1s_tests.txt
for c in cfg3 cfg5; do
  timeout 600 python bench.py --config $c --no-cpu-baseline > $OUT/bench_$c.json 2>> $OUT/bench.err
done
timeout 600 python bench.py --no-cpu-baseline --no-train-object > $OUT/bench_cfg2.json 2>> $OUT/bench.err
cat $OUT/k1s_tests.txt
