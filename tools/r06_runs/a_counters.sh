#!/bin/bash
# round 6, item 2: SQ / TCP / TCC counters of the persistent kernels (c5: 512 threads, c2: 256 threads), the pair kernels (c4) and c3
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06
timeout 3000 python3 tools/persistent_counters.py --out gpurun_out/r06/fused3_counters.log c5 c2 c4 c3 > gpurun_out/r06/fused3_counters.stdout 2>&1
echo rc=$?
