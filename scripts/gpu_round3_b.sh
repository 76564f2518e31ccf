#!/bin/bash
# full-size tests + which of the 64 sweep items trip the a-posteriori guard
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu_b.log | tail -15
timeout -k 10 600 python3 scripts/sweep_guard_report.py -v 2>&1 | tee gpurun_out/sweep_guard.log | tail -70
