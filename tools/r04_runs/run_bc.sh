#!/bin/bash
# round 4, run bc: seeded random soak of the public API after the split-complex work (2000 cases up to 2^21 points, then 600 up to 2^23)
mkdir -p gpurun_out/r04bc
PYFFT_AMD_SWEEP=2000:4242:21 timeout 2400 python -m pytest tests/test_random_sweep_gpu.py -q -x 2>&1 | tail -4 > gpurun_out/r04bc/soak1.log
PYFFT_AMD_SWEEP=600:9191:23 timeout 2400 python -m pytest tests/test_random_sweep_gpu.py -q -x 2>&1 | tail -4 > gpurun_out/r04bc/soak2.log
cat gpurun_out/r04bc/soak1.log gpurun_out/r04bc/soak2.log
