#!/bin/bash
# full GPU suite, smoke, default bench, 2-rank rehearsal on one GPU (gloo, both ranks on device 0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py > gpurun_out/bench_full.log 2>&1; grep '^{' gpurun_out/bench_full.log | cut -c1-330
PLFEM_BENCH_BACKEND=gloo PLFEM_BENCH_SAME_DEVICE=1 python3 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline 2>gpurun_out/two_rank.log | cut -c1-300
PLFEM_BENCH_BACKEND=gloo PLFEM_BENCH_SAME_DEVICE=1 python3 bench.py --gpus 2 --sweep --steps 1 --warmup 0 2>>gpurun_out/two_rank.log | cut -c1-300
tail -3 gpurun_out/two_rank.log
