#!/bin/bash
mkdir -p gpurun_out/r04aq
timeout 900 python tools/fused_sweep.py 1048576 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1,auto@MIFFT_NARROW_TILES=2 1024x1024 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 \
  524288 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 1024x512 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 512x1024 float32 0.125 auto,auto@MIFFT_NARROW_TILES=1 \
  1048576 float32 0.0625 auto,auto@MIFFT_NARROW_TILES=1 > gpurun_out/r04aq/sweep3.log 2>&1
cat gpurun_out/r04aq/sweep3.log
