# One steady-state training step under rocprofv3: the kernel families (tools/kernel_families.py) and the launch sequence
# (tools/step_sequence.py).  usage: bash tools/trace_step_sequence.sh [out-dir under gpurun_out] [extra bench.py flags]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-seq}
shift
mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/tr -o t -- python bench.py --train --steps 3 --warmup 2 "$@" > $out/b.json 2> $out/err.txt
f=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python tools/step_sequence.py $f > $out/sequence.txt
python tools/kernel_families.py $f 80 > $out/families.txt
rm -rf $out/tr
