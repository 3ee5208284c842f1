#!/bin/bash
# round 4, run az: plane passes of split-complex multi-pass plans on the tiled fixed-shape kernel (planes in, interleaved out)
mkdir -p gpurun_out/r04at
timeout 1500 python -m pytest tests -m gpu -q -x -k "float32 or float64 or split or f32 or f64 or random or grid or golden" 2>&1 | tail -5 > gpurun_out/r04at/tests6.log
timeout 900 python tools/fused_sweep.py 128x128x128 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 64x64x64 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 32x32x128 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 \
  128x128x128 float64 0.125 auto,auto@MIFFT_NARROW_TILES=1 64x64x64 float64 0.125 auto,auto@MIFFT_NARROW_TILES=1 128x128x128 float32 0.03125 auto,auto@MIFFT_NARROW_TILES=1 \
  16x16x128 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 256x64x64 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 > gpurun_out/r04at/nd4.log 2>&1
cat gpurun_out/r04at/tests6.log; cat gpurun_out/r04at/nd4.log
