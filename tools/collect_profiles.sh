#!/bin/bash
# Round profiles on the GPU box: kernel-trace stats of a bench run, then SEPARATE --pmc passes
# (gpurun refuses --pmc combined with trace domains other than --kernel-trace/--stats) for the HBM
# traffic of the encoder group and of the extract+paste pair and for the SQ MFMA-busy counters.
# Everything lands in gpurun_out/$1/ named $2_* (default r04); the summaries to commit are copied into
# profiles/ by hand.  Every command runs under `timeout`.
set -u
OUT=gpurun_out/${1:-prof}
R3=${2:-r06}
mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
R="timeout 900 rocprofv3 --output-format csv"
B="--no-cpu-baseline --no-train-object"
# kernel durations that can be compared with the bench line's roofline: one batch in flight (with 4 in
# flight the kernels of different batches share the chip and every duration is inflated by the overlap);
# the second trace is the default run, for the record
$R --kernel-trace --stats -d $OUT/trace -o t -- python bench.py --in-flight 1 --steps 5 --warmup 2 $B > $OUT/trace_bench.json 2> $OUT/trace.log
$R --kernel-trace --stats -d $OUT/trace4 -o t -- python bench.py --steps 8 --warmup 4 $B > $OUT/trace4_bench.json 2> $OUT/trace4.log
for which in enc attn; do
  $R --pmc FETCH_SIZE -d $OUT/pmc_${which}_f -o f -- python bench.py --pmc-group 10 --pmc-which $which > $OUT/pmc_${which}_f.json 2> $OUT/pmc_${which}_f.log
  $R --pmc WRITE_SIZE -d $OUT/pmc_${which}_w -o w -- python bench.py --pmc-group 10 --pmc-which $which > $OUT/pmc_${which}_w.json 2> $OUT/pmc_${which}_w.log
done
$R --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 -d $OUT/pmc_sq -o s -- python bench.py --pmc-group 10 --pmc-which enc > $OUT/pmc_sq.json 2> $OUT/pmc_sq.log
F=$(find $OUT/pmc_enc_f -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_enc_w -name '*counter_collection.csv' | head -1)
timeout 60 python tools/pmc_traffic.py $F $W 10 8 512 $OUT/${R3}_pmc_encoder_traffic.json > $OUT/traffic_enc.txt 2>&1
F=$(find $OUT/pmc_attn_f -name '*counter_collection.csv' | head -1); W=$(find $OUT/pmc_attn_w -name '*counter_collection.csv' | head -1)
timeout 60 python tools/pmc_traffic.py $F $W 10 8 512 $OUT/${R3}_pmc_attn_traffic.json "ra::attnd::" > $OUT/traffic_attn.txt 2>&1
S=$(find $OUT/pmc_sq -name '*counter_collection.csv' | head -1)
timeout 60 python tools/pmc_summary.py $S > $OUT/${R3}_pmc_sq_mfma_per_kernel.csv 2> $OUT/sq.txt
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/${R3}_bench_kernel_stats.csv
cp $(find $OUT/trace4 -name '*kernel_stats.csv' | head -1) $OUT/${R3}_bench_kernel_stats_pipeline.csv
# keep the merge small: the raw traces are not needed
rm -rf $OUT/trace $OUT/trace4 $OUT/pmc_*_f $OUT/pmc_*_w $OUT/pmc_sq
# the traffic files must be in profiles/ for the bench line to pick them up
cp $OUT/${R3}_pmc_encoder_traffic.json $OUT/${R3}_pmc_attn_traffic.json $OUT/${R3}_pmc_sq_mfma_per_kernel.csv profiles/ 2>/dev/null  # (roofline.traffic / mfma_busy / executed read them)
timeout 900 python bench.py > $OUT/${R3}_bench_n1.json 2> $OUT/bench.err
for DT in f32 bf16; do
  SUF=""; [ $DT = bf16 ] && SUF="_bf16"
  timeout 900 python bench.py --train --dtype $DT --steps 4 --warmup 3 > $OUT/${R3}_train${SUF}_n1.json 2>> $OUT/bench.err
  $R --kernel-trace --stats -d $OUT/trace_train -o t -- python bench.py --train --dtype $DT --steps 4 --warmup 2 > $OUT/trace_train.json 2> $OUT/trace_train.log
  cp $(find $OUT/trace_train -name '*kernel_stats.csv' | head -1) $OUT/${R3}_train${SUF}_kernel_stats.csv
  # one steady-state step out of the trace (the stats file above also holds model set-up and the eager first step)
  timeout 120 python tools/kernel_families.py $(find $OUT/trace_train -name '*kernel_trace.csv' | head -1) 30 > $OUT/${R3}_train${SUF}_kernel_families.txt
  rm -rf $OUT/trace_train
done
timeout 900 python bench.py --config cfg3 > $OUT/${R3}_bench_cfg3.json 2>> $OUT/bench.err
timeout 900 python bench.py --config cfg5 > $OUT/${R3}_bench_cfg5.json 2>> $OUT/bench.err
timeout 120 tools/bin/attn_probe 8 > $OUT/${R3}_attn_probe.txt 2>&1
{ timeout 120 tools/bin/pair8_probe 8; timeout 120 tools/bin/pair8_probe 16; [ -x tools/bin/pairw_probe ] && { timeout 120 tools/bin/pairw_probe 8; timeout 120 tools/bin/pairw_probe 16; }; } > $OUT/${R3}_pair_probes.txt 2>&1
timeout 120 tools/bin/lat_probe 256 > $OUT/${R3}_lat_probe.txt 2>&1
# round 4: K1 at the training step's full-resolution shapes, the Hungarian launch problem by problem, the one-XCD barrier
{ for shp in "8 8 512 512 8" "4 8 512 512 8" "16 16 256 256 8" "32 32 128 128 8"; do echo "== Cin Cout H W B = $shp"; timeout 120 python tools/conv_shape_bench.py $shp 2>&1 | grep " us "; done; } > $OUT/${R3}_conv_shape_bench.txt
timeout 300 python tools/hungarian_step_probe.py 2>&1 | tail -3 > $OUT/${R3}_hungarian_step_probe.txt
[ -x tools/bin/xcd_barrier_probe ] && { timeout 60 tools/bin/xcd_barrier_probe 48; timeout 60 tools/bin/xcd_barrier_probe 16; } > $OUT/${R3}_xcd_barrier_probe.txt 2>&1

# round 5, second session: what the pipeline's slots do to each other (static tile walk / drawn tiles), per launch and as a group
{ timeout 200 python tools/contention_probe.py 16; RA_ENGINE_TICKETS=1 timeout 200 python tools/contention_probe.py 16; timeout 250 python tools/contention_by_layer.py 16; } 2>&1 | grep -v amdgpu.ids > $OUT/${R3}_contention_probes.txt
timeout 120 python tools/ctrl_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/${R3}_ctrl_bench.txt
RA_CTRL_XCD=0 timeout 120 python tools/ctrl_bench.py 2>&1 | grep -v amdgpu.ids | sed 's/^/RA_CTRL_XCD=0  /' >> $OUT/${R3}_ctrl_bench.txt
[ -x tools/bin/mfma_rate_probe ] && timeout 60 tools/bin/mfma_rate_probe > $OUT/${R3}_mfma_rate_probe.txt 2>&1
ls -la $OUT
