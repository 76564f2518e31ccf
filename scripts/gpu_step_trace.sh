#!/bin/bash
# kernel trace of a short bench run, summarised three ways (run through gpurun): the timeline of one block Lanczos
# step, the per-level table of the sweeps and the per-level table of the factorisation
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_step
rocprofv3 --kernel-trace -d gpurun_out/prof_step -o st --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_st.log 2>&1 || exit 1
python3 scripts/step_timeline.py gpurun_out/prof_step/st_kernel_trace.csv 70 > gpurun_out/step_timeline.txt
python3 scripts/level_roofline.py gpurun_out/prof_step/st_kernel_trace.csv > gpurun_out/levels_solve.txt
python3 scripts/factor_levels.py gpurun_out/prof_step/st_kernel_trace.csv > gpurun_out/levels_factor.txt
rm -rf gpurun_out/prof_step
cat gpurun_out/levels_factor.txt
