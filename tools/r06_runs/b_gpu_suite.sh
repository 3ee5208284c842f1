#!/bin/bash
# round 6: the whole GPU suite with the soak-only cases un-hidden (timing + the 25 slowest)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06
timeout 2400 python -m pytest tests -m gpu -q -x --durations=30 > gpurun_out/r06/gpu_suite.log 2>&1
echo rc=$? >> gpurun_out/r06/gpu_suite.log
tail -45 gpurun_out/r06/gpu_suite.log
