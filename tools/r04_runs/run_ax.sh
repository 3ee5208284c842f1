#!/bin/bash
# round 4, run ax: the reference's published shapes, interleaved against split planes
mkdir -p gpurun_out/r04at
timeout 900 python tools/perf_table.py --split > gpurun_out/r04at/perf_split.log 2>&1
cat gpurun_out/r04at/perf_split.log
