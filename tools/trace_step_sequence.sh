cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/seq
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seq/tr -o t -- python bench.py --train --steps 3 --warmup 2 > gpurun_out/seq/b.json 2> gpurun_out/seq/err.txt
f=$(find gpurun_out/seq/tr -name '*kernel_trace.csv' | head -1)
python tools/step_sequence.py $f > gpurun_out/seq/sequence.txt
python tools/kernel_families.py $f 80 > gpurun_out/seq/families.txt
rm -rf gpurun_out/seq/tr
tail -3 gpurun_out/seq/b.json | cut -c1-300
