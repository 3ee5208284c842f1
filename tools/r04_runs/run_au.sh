#!/bin/bash
# round 4, run au: split-complex rows on the register-edged kernels
mkdir -p gpurun_out/r04at
timeout 1500 python -m pytest tests -m gpu -q -x -k "float32 or float64 or split or f32 or f64 or random or grid or golden" 2>&1 | tail -5 > gpurun_out/r04at/tests.log
timeout 900 python tools/fused_sweep.py 1024 float32 1 auto,auto@MIFFT_NARROW_TILES=1 4096 float32 1 auto,auto@MIFFT_NARROW_TILES=1 8192 float32 1 auto,auto@MIFFT_NARROW_TILES=1 32768 float32 1 auto,auto@MIFFT_NARROW_TILES=1 \
  256 float32 1 auto,auto@MIFFT_NARROW_TILES=1 16384 float32 1 auto,auto@MIFFT_NARROW_TILES=1 1024 float64 1 auto,auto@MIFFT_NARROW_TILES=1 4096 float64 1 auto,auto@MIFFT_NARROW_TILES=1 16384 float64 1 auto,auto@MIFFT_NARROW_TILES=1 \
  1024 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 8192 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 1024x1024 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 1024x1024 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 > gpurun_out/r04at/rows2.log 2>&1
cat gpurun_out/r04at/tests.log; cat gpurun_out/r04at/rows2.log
