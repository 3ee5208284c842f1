#!/bin/bash
mkdir -p gpurun_out/r04at
timeout 900 python tools/perf_table.py --split > gpurun_out/r04at/perf_split3.log 2>&1
cat gpurun_out/r04at/perf_split3.log
