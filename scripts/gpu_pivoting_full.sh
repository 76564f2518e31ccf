#!/bin/bash
# the complete list of random cross-sections of tests/test_gpu_pivoting.py (104 vectorial + 24 scalar), summary into
# gpurun_out/r03_pivoting_full.txt (committed as profiles/r03_pivoting_full.txt); then the plain -m gpu suite, timed
set -e -o pipefail
mkdir -p gpurun_out
export PLFEM_PIVOT_CASES=104 PLFEM_PIVOT_CASES_SCALAR=24
timeout -k 10 900 python3 -m pytest tests/test_gpu_pivoting.py -m gpu -x -q -s 2>&1 | tee gpurun_out/pivoting_full.log | grep -E "chunk|scalar pencil|passed|failed"
grep -E "chunk|scalar pencil|passed|failed" gpurun_out/pivoting_full.log > gpurun_out/r03_pivoting_full.txt
unset PLFEM_PIVOT_CASES PLFEM_PIVOT_CASES_SCALAR
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --durations=8 2>&1 | tee gpurun_out/pytest_gpu.log | tail -14
