#!/bin/bash
# round 4, run av: split-complex single-launch N-D plans through the tiled fixed-shape kernel
mkdir -p gpurun_out/r04at
timeout 1500 python -m pytest tests -m gpu -q -x -k "float32 or float64 or split or f32 or f64 or random or grid or golden" 2>&1 | tail -5 > gpurun_out/r04at/tests3.log
timeout 900 python tools/fused_sweep.py 128x128 float32 1 auto,auto@MIFFT_NARROW_TILES=1 64x64 float32 1 auto,auto@MIFFT_NARROW_TILES=1 16x16 float32 1 auto,auto@MIFFT_NARROW_TILES=1 16x16x16 float32 1 auto,auto@MIFFT_NARROW_TILES=1 \
   32x32x32 float32 1 auto,auto@MIFFT_NARROW_TILES=1 128x128 float64 1 auto,auto@MIFFT_NARROW_TILES=1 16x16x16 float64 1 auto,auto@MIFFT_NARROW_TILES=1 128x128 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 \
   16x16x16 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 16x16 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 > gpurun_out/r04at/nd.log 2>&1
cat gpurun_out/r04at/tests3.log; cat gpurun_out/r04at/nd.log
