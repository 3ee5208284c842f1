timeout 600 python -m pytest tests/test_errors_gpu.py -x -q -k "fused" 2>&1 | tail -5
for lr in 6,14 5,14 7,14 4,14 8,14 6,12 3,14; do
echo "== $lr"; PYFFT_AMD_FUSED3=$lr timeout 300 python tools/quick_bench.py one 1048576 complex128 512 | tail -1
done
PYFFT_AMD_FUSED3=6,14 timeout 300 python tools/quick_bench.py one 1048576 complex128 512 inplace | tail -1
PYFFT_AMD_FUSED3=6,14 timeout 300 python tools/quick_bench.py one 1048576 float64 512 | tail -1
PYFFT_AMD_STRATEGY=pipelined timeout 300 python tools/quick_bench.py one 1048576 complex128 512 | tail -1
PYFFT_AMD_STRATEGY=pipelined timeout 300 python tools/quick_bench.py one 1048576 float64 512 | tail -1
PYFFT_AMD_STRATEGY=pipelined timeout 300 python tools/quick_bench.py one 1024x1024 complex128 512 | tail -1
