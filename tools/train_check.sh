mkdir -p gpurun_out/dbg
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_bf16_storage_gpu.py tests/test_train_small_gpu.py -q 2>&1 | tail -3
timeout 1800 python -m pytest tests/test_train_gpu.py -q -x > gpurun_out/dbg/train.txt 2>&1; tail -3 gpurun_out/dbg/train.txt
python bench.py --train --steps 4 --warmup 3 2>&1 | tail -1 | cut -c1-330
python bench.py --train --dtype bf16 --steps 4 --warmup 3 2>&1 | tail -1 | cut -c1-330
bash tools/trace_step_sequence.sh seq > /dev/null 2>&1
head -12 gpurun_out/seq/families.txt
