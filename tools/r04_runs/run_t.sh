#!/bin/bash
# round 4, run t: split-plane tiled batches (tests), the survey of shapes still on two / three launches per chunk
mkdir -p gpurun_out/r04t
python -m pytest tests/test_round4_gpu.py -q -x -k "tiled_batch_split" 2>&1 | tail -5 > gpurun_out/r04t/tests.log
python tools/quick_bench.py tail 2>&1 | sed 's/passes=\[.*\]//' > gpurun_out/r04t/tail.log
python tools/generic_probe.py tiled 2>&1 | tail -30 > gpurun_out/r04t/tiled.log
cat gpurun_out/r04t/tests.log
