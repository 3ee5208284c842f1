#!/bin/bash
# round 4, run ay: write-through stores in the run-time-shaped N-D kernel (small split-complex launches); the reference's shapes again
mkdir -p gpurun_out/r04at
timeout 1500 python -m pytest tests -m gpu -q -x -k "float32 or float64 or split or f32 or f64 or random or grid or golden or small" 2>&1 | tail -5 > gpurun_out/r04at/tests5.log
timeout 900 python tools/fused_sweep.py 16x16 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 16x16x16 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 8x8x64 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 \
  16x16 float64 0.03125 auto,auto@MIFFT_NARROW_TILES=1 16x16x16 float64 0.03125 auto,auto@MIFFT_NARROW_TILES=1 16 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 > gpurun_out/r04at/nd3.log 2>&1
timeout 900 python tools/perf_table.py --split > gpurun_out/r04at/perf_split2.log 2>&1
cat gpurun_out/r04at/tests5.log; cat gpurun_out/r04at/nd3.log; head -13 gpurun_out/r04at/perf_split2.log
