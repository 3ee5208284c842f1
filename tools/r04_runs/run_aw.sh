#!/bin/bash
# round 4, run aw: big split-complex N-D shapes in one launch
mkdir -p gpurun_out/r04at
timeout 1500 python -m pytest tests -m gpu -q -x -k "float32 or float64 or split or f32 or f64 or random or grid or golden" 2>&1 | tail -5 > gpurun_out/r04at/tests4.log
timeout 900 python tools/fused_sweep.py 32x32x32 float32 1 auto,auto@MIFFT_NARROW_TILES=1 128x128 float64 1 auto,auto@MIFFT_NARROW_TILES=1 16x32x32 float64 1 auto,auto@MIFFT_NARROW_TILES=1 64x128 float32 1 auto,auto@MIFFT_NARROW_TILES=1 \
   128x128 float32 1 auto 8x32x32 float32 1 auto,auto@MIFFT_NARROW_TILES=1 32x32x32 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 > gpurun_out/r04at/nd2.log 2>&1
cat gpurun_out/r04at/tests4.log; cat gpurun_out/r04at/nd2.log
