#!/bin/bash
# step timeline + context-creation trace + host analysis trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/prof_step
rocprofv3 --kernel-trace -d gpurun_out/prof_step -o st --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_st.log 2>&1 || exit 1
python3 scripts/step_timeline.py gpurun_out/prof_step/st_kernel_trace.csv 70 > gpurun_out/step_timeline.txt
rm -rf gpurun_out/prof_step
cat gpurun_out/step_timeline.txt
PLFEM_CTX_TRACE=1 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | grep "^\[ctx\]" | tail -4
