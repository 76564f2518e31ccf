#!/bin/bash
# sweep kernels after a change: the parity tests that exercise them, then the per-level table and the step timeline
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scalar.py -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu_f.log | tail -5
bash scripts/gpu_step_trace.sh > /dev/null
cat gpurun_out/levels_solve.txt
tail -3 gpurun_out/step_timeline.txt
grep -E "^total" gpurun_out/levels_factor.txt
