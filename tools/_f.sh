timeout 900 python -m pytest tests/test_errors_gpu.py tests/test_round2_gpu.py -x -q -k "fused or split_plane or mailbox or async" 2>&1 | tail -2
timeout 300 python bench.py --plain --steps 5 --warmup 2 | tail -1 | cut -c1-200
timeout 300 python bench.py --plain --config c5 --steps 5 --warmup 2 | tail -1 | cut -c1-200
timeout 300 python tools/quick_bench.py one 2097152 complex64 512 | tail -1 | cut -c100-220
timeout 300 python tools/quick_bench.py one 1048576 complex128 512 | tail -1 | cut -c100-220
timeout 300 python tools/quick_bench.py one 1048576 float32 4096 | tail -1 | cut -c100-220
