#!/bin/bash
# the complete list of random cross-sections of tests/test_gpu_pivoting.py (104 vectorial + 24 scalar), summary into
# gpurun_out/<tag>_pivoting_full.txt (committed as profiles/<tag>_pivoting_full.txt)
# usage: gpu_pivoting_full.sh [tag]   (default r04)
set -o pipefail
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PLFEM_PIVOT_CASES=104 PLFEM_PIVOT_CASES_SCALAR=24
timeout -k 10 1000 python3 -m pytest tests/test_gpu_pivoting.py -m gpu -q -s 2>&1 | tee gpurun_out/pivoting_full.log | grep -E "chunk|scalar pencil|singular|passed|failed"
grep -E "chunk|scalar pencil|singular|passed|failed" gpurun_out/pivoting_full.log > gpurun_out/${TAG}_pivoting_full.txt
