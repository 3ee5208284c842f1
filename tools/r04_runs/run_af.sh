#!/bin/bash
# round 4, run af: soak of the persistent strategies against the chain, whole arrays
mkdir -p gpurun_out/r04af
timeout 2400 python tools/persistent_soak.py 5 > gpurun_out/r04af/soak.log 2>&1
tail -12 gpurun_out/r04af/soak.log; grep -c " ok$" gpurun_out/r04af/soak.log; grep MISMATCH gpurun_out/r04af/soak.log | head
