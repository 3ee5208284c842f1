#!/bin/bash
mkdir -p gpurun_out/r04at
timeout 1500 python -m pytest tests -m gpu -q -x -k "float64 or split or f64 or random or grid or golden or pair" 2>&1 | tail -5 > gpurun_out/r04at/tests7.log
timeout 900 python tools/fused_sweep.py 128x128x128 float64 0.125 auto 64x128x128 float64 0.125 auto 128x128x128 float64 2 auto 128x128x128 float64 0.03125 auto > gpurun_out/r04at/nd5.log 2>&1
cat gpurun_out/r04at/tests7.log; cat gpurun_out/r04at/nd5.log
