#!/bin/bash
# round 4, run v: L2 <-> fabric traffic of the split-complex forms of configs 2 and 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04v
timeout 800 python3 tools/pmc_traffic.py --tag r04v c2s c3s > gpurun_out/r04v/pmc_traffic.log 2>&1
cp profiles/traffic_c2s.json profiles/traffic_c3s.json profiles/r04v_* gpurun_out/r04v/ 2>/dev/null
tail -5 gpurun_out/r04v/pmc_traffic.log
