python tools/conv_shape_bench.py 8 8 512 512 8 2>&1 | grep "us "
echo "--- 16->16 at 256"; python tools/conv_shape_bench.py 16 16 256 256 8 2>&1 | grep "us "
echo "--- 4->8 at 512"; python tools/conv_shape_bench.py 4 8 512 512 8 2>&1 | grep "us "
echo "--- 32->32 at 128"; python tools/conv_shape_bench.py 32 32 128 128 8 2>&1 | grep "us "
timeout 600 python -m pytest tests/test_bf16_storage_gpu.py tests/test_kernels_gpu.py -q 2>&1 | tail -2
