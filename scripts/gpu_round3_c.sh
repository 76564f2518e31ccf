#!/bin/bash
# new tests of round 3 (pivoting, golden fixtures, interface KAT, C1 vs oracle)
set -e -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests/test_gpu_pivoting.py -m gpu -x -v -s --durations=8 2>&1 | tee gpurun_out/pytest_gpu_c.log
