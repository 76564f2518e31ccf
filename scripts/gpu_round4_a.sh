#!/bin/bash
# Round-4 host-side baseline: where the cold step's wall time goes (python phases, symbolic sub-phases, context trace).
set -e -o pipefail
mkdir -p gpurun_out/r4a
export PLFEM_MALLOC_TUNE=1
timeout -k 10 300 python3 scripts/profile_python.py > gpurun_out/r4a/profile_python.txt 2>&1
timeout -k 10 120 python3 scripts/time_symbolic.py 1 > gpurun_out/r4a/time_symbolic.txt 2>&1
PLFEM_CTX_TRACE=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/r4a/bench_ctx_trace.log >gpurun_out/r4a/bench_short.json
tail -4 gpurun_out/r4a/profile_python.txt
tail -12 gpurun_out/r4a/time_symbolic.txt
tail -4 gpurun_out/r4a/bench_ctx_trace.log
cut -c1-1200 gpurun_out/r4a/bench_short.json
