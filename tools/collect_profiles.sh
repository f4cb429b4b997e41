#!/bin/bash
# Round profiles on the GPU box: kernel-trace stats of a bench run, then SEPARATE --pmc passes
# (gpurun refuses --pmc combined with trace domains other than --kernel-trace/--stats) for the HBM
# traffic of the encoder group and of the extract+paste pair and for the SQ MFMA-busy counters.
# Everything lands in gpurun_out/$1/; the summaries to commit are copied into profiles/ by hand.
set -u
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
R="rocprofv3 --output-format csv"
# kernel durations that can be compared with the bench line's roofline: one batch in flight (with 4 in
# flight the kernels of different batches share the chip and every duration is inflated by the overlap);
# the second trace is the default run, for the record
$R --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --in-flight 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.log
$R --kernel-trace --stats -d $OUT/trace4 -o t -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline > $OUT/trace4_bench.json 2> $OUT/trace4.log
for which in enc attn; do
  $R --pmc FETCH_SIZE -d $OUT/pmc_${which}_f -o f -- python bench.py --pmc-group 10 --pmc-which $which > $OUT/pmc_${which}_f.json 2> $OUT/pmc_${which}_f.log
  $R --pmc WRITE_SIZE -d $OUT/pmc_${which}_w -o w -- python bench.py --pmc-group 10 --pmc-which $which > $OUT/pmc_${which}_w.json 2> $OUT/pmc_${which}_w.log
done
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o s -- python bench.py --pmc-group 10 --pmc-which enc > $OUT/pmc_sq.json 2> $OUT/pmc_sq.log
F=$(find $OUT/pmc_enc_f -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_enc_w -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $F $W 10 8 512 $OUT/r02_pmc_encoder_traffic.json > $OUT/traffic_enc.txt 2>&1
F=$(find $OUT/pmc_attn_f -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_attn_w -name '*counter_collection.csv' | head -1)
python tools/pmc_traffic.py $F $W 10 8 512 $OUT/r02_pmc_attn_traffic.json "ra::attnd::" > $OUT/traffic_attn.txt 2>&1
S=$(find $OUT/pmc_sq -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py $S > $OUT/r02_pmc_sq_mfma_per_kernel.csv 2> $OUT/sq.txt
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/r02_bench_kernel_stats.csv
cp $(find $OUT/trace4 -name '*kernel_stats.csv' | head -1) $OUT/r02_bench_kernel_stats_pipeline.csv
# keep the merge small: the raw traces are not needed
rm -rf $OUT/trace $OUT/trace4 $OUT/pmc_*_f $OUT/pmc_*_w $OUT/pmc_sq
python bench.py --attn-b32 > $OUT/r02_bench_n1.json 2> $OUT/bench.err
python bench.py --train --steps 4 --warmup 3 > $OUT/r02_train_n1.json 2>> $OUT/bench.err
$R --kernel-trace --stats -d $OUT/trace_train -o t -- python bench.py --train --steps 4 --warmup 2 > $OUT/trace_train.json 2> $OUT/trace_train.log
cp $(find $OUT/trace_train -name '*kernel_stats.csv' | head -1) $OUT/r02_train_kernel_stats.csv
python tools/kernel_families.py $OUT/r02_train_kernel_stats.csv 6 > $OUT/r02_train_kernel_families.txt
rm -rf $OUT/trace_train
python bench.py --config cfg3 > $OUT/r02_bench_cfg3.json 2>> $OUT/bench.err
python bench.py --config cfg5 > $OUT/r02_bench_cfg5.json 2>> $OUT/bench.err
ls -la $OUT
