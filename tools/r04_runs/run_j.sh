#!/bin/bash
# round 4, call j: dense smooth rows on the single-buffer tile kernel against the two-buffer row kernel; 3-D smooth shapes as planes + z lines
set -u
OUT=gpurun_out/r04j
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python3 -m pytest tests/test_round4_gpu.py tests/test_round2_gpu.py tests/test_round3_gpu.py -q -m gpu -k "smooth or any_size or mixed or generic" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
MIFFT_ROWS_ND=2 timeout 900 python3 -m pytest tests/test_round2_gpu.py tests/test_round3_gpu.py -q -m gpu -k "any_size or mixed" > $OUT/pytest_nd_rows.log 2>&1; tail -5 $OUT/pytest_nd_rows.log
PROBE='
import sys; sys.path.insert(0,"."); sys.path.insert(0,"tools")
import numpy
from generic_probe import run
for n in (12, 60, 100, 120, 240, 360, 500, 1000, 1500, 2000, 2187, 3000, 3125, 4000, 4096, 6000, 8000, 10000, 12000, 15000):
    try:
        run((n,), numpy.complex64, (1<<27)//n)
    except Exception as e:
        print(n, "FAILED", repr(e)[:100])
for n in (100, 1000, 2000, 4000, 5000):
    run((n,), numpy.complex128, (1<<26)//n)
'
echo "# two-buffer row kernel (MIFFT_ROWS_ND=1)" > $OUT/rows_ab.log
MIFFT_ROWS_ND=1 python3 -c "$PROBE" >> $OUT/rows_ab.log 2>&1
echo "# single-buffer tile kernel wherever it fits (MIFFT_ROWS_ND=2)" >> $OUT/rows_ab.log
MIFFT_ROWS_ND=2 python3 -c "$PROBE" >> $OUT/rows_ab.log 2>&1
cat $OUT/rows_ab.log
python3 -c '
import sys; sys.path.insert(0,"."); sys.path.insert(0,"tools")
import numpy
from generic_probe import run
run((60, 60, 60), numpy.complex64, 512); run((100, 100, 100), numpy.complex64, 64); run((30, 60, 120), numpy.complex128, 128)
' > $OUT/planes.log 2>&1; cat $OUT/planes.log
