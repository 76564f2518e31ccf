#!/bin/bash
# experiment driver: GPU parity tests, then a kernel trace of bench.py summarised per tree level
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -30 gpurun_out/pytest_gpu.log; exit 1; }
rm -rf gpurun_out/prof_lv
rocprofv3 --kernel-trace -d gpurun_out/prof_lv -o lv --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_lv.log 2>&1 || exit 1
python3 scripts/level_roofline.py gpurun_out/prof_lv/lv_kernel_trace.csv > gpurun_out/levels_solve.txt
python3 scripts/factor_levels.py gpurun_out/prof_lv/lv_kernel_trace.csv > gpurun_out/levels_factor.txt
python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench.log 2>&1
tail -2 gpurun_out/pytest_gpu.log; grep sweep gpurun_out/levels_solve.txt; cat gpurun_out/levels_factor.txt; grep "^{" gpurun_out/bench.log | cut -c1-200
