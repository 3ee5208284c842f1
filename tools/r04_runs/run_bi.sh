#!/bin/bash
# round 4, run bi: write-through quad stores of planes in the LDS-staged tile kernel (small launches); split tests; the reference's shapes
mkdir -p gpurun_out/r04bi
timeout 1500 python -m pytest tests -m gpu -q -x -k "float32 or float64 or split or f32 or f64 or random or grid or golden or small" 2>&1 | tail -4 > gpurun_out/r04bi/tests.log
timeout 900 python tools/perf_table.py --split > gpurun_out/r04bi/perf_split.log 2>&1
cat gpurun_out/r04bi/tests.log; head -13 gpurun_out/r04bi/perf_split.log
