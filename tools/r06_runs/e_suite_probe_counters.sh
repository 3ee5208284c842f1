#!/bin/bash
# round 6: the whole GPU suite with the coverage cases, the three-launch composite probe (review item 6), counters of c5 / c2 after the look-ahead
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r06; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x --durations=40 > $O/e_suite.log 2>&1; echo "suite rc=$?" >> $O/e_suite.log; tail -50 $O/e_suite.log
timeout 900 python3 tools/three_launch_probe.py 2 > $O/e_three_launch.log 2>&1; cat $O/e_three_launch.log
timeout 1500 python3 tools/persistent_counters.py --out $O/e_counters_after.log c5 c2 > $O/e_counters_after.stdout 2>&1; tail -45 $O/e_counters_after.log
