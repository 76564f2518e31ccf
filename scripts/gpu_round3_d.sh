#!/bin/bash
# one launch per block step (column workgroups), X in the upper triangle until the end: the tests that read the fronts
# back, the full-size cases, then a short bench and the per-level table of the factorisation
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scalar.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu_d.log | tail -8
bash scripts/gpu_step_trace.sh | tail -22
