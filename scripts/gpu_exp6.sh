#!/bin/bash
# forward mixed-form threshold sweep: per-solve sweep time from plfem_debug_solve_block under a kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for thr in 64 96 128 160 192 100000; do
  rm -rf gpurun_out/prof_f
  PLFEM_MIX_BIG_S2=$thr rocprofv3 --kernel-trace -d gpurun_out/prof_f -o f --output-format csv -- python3 scripts/sweep_filter_timing.py > gpurun_out/filter.log 2>&1 || { tail -5 gpurun_out/filter.log; exit 1; }
  echo "threshold $thr"; python3 scripts/sweep_filter_table.py gpurun_out/prof_f/f_kernel_trace.csv | awk '{print $1, $2, $3}' | head -14 | tr '\n' ';'; echo; python3 scripts/sweep_filter_table.py gpurun_out/prof_f/f_kernel_trace.csv | tail -1
done
rm -rf gpurun_out/prof_f
