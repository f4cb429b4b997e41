set -u
export TMPDIR=/tmp
O=gpurun_out/r3t; mkdir -p $O
timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -o t -- python bench.py --train --steps 4 --warmup 2 > $O/trace_train.json 2> $O/trace.log
F=$(find $O/trace -name '*kernel_stats.csv' | head -1)
cp $F $O/train_kernel_stats.csv
python tools/kernel_families.py $O/train_kernel_stats.csv 6 40 > $O/families.txt
rm -rf $O/trace
cat $O/families.txt
