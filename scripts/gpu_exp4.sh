#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -rf gpurun_out/prof_f
rocprofv3 --kernel-trace -d gpurun_out/prof_f -o f --output-format csv -- python3 scripts/sweep_filter_timing.py > gpurun_out/filter.log 2>&1 || { tail -5 gpurun_out/filter.log; exit 1; }
python3 scripts/sweep_filter_table.py gpurun_out/prof_f/f_kernel_trace.csv | tee gpurun_out/sweep_filter.txt
rm -rf gpurun_out/prof_f
PLFEM_CTX_TRACE=1 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>gpurun_out/ctx.log | cut -c1-400
grep "^\[ctx\]" gpurun_out/ctx.log | tail -2
