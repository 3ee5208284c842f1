#!/bin/bash
# round 4, run ad: split-complex fp32 2^16 ... 2^18: sibling tiles on the global list against the per-XCD lists, same process
mkdir -p gpurun_out/r04ac
timeout 900 python tools/fused_sweep.py 65536 float32 2 auto,auto@PYFFT_AMD_SPLIT_FUSEDX=1,f:112:224,auto,auto@PYFFT_AMD_SPLIT_FUSEDX=1 \
   131072 float32 2 auto,auto@PYFFT_AMD_SPLIT_FUSEDX=1,auto,auto@PYFFT_AMD_SPLIT_FUSEDX=1 262144 float32 2 auto,auto@PYFFT_AMD_SPLIT_FUSEDX=1,f:14:28,auto,auto@PYFFT_AMD_SPLIT_FUSEDX=1 \
   65536 float32 0.5 auto,auto@PYFFT_AMD_SPLIT_FUSEDX=1 262144 float32 0.5 auto,auto@PYFFT_AMD_SPLIT_FUSEDX=1 > gpurun_out/r04ac/sweep2.log 2>&1
tail -40 gpurun_out/r04ac/sweep2.log
