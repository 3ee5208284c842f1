timeout 900 python -m pytest tests/test_errors_gpu.py -x -q -k "fused" 2>&1 | tail -8
timeout 300 python bench.py --plain --config c3 --steps 5 --warmup 2 | tail -1 | cut -c1-260
timeout 300 python tools/quick_bench.py one 1024x1024 complex64 512 inplace | tail -1 | cut -c60-220
timeout 300 python tools/quick_bench.py one 1024x1024 float32 512 | tail -1 | cut -c60-220
PYFFT_AMD_STRATEGY=pipelined timeout 300 python tools/quick_bench.py one 1024x1024 float32 512 | tail -1 | cut -c60-220
