#!/bin/bash
# column workgroups per front on the throughput levels, lean / fused column block: A/B by environment
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "1 -1" "2 -1" "4 -1" "1 0" "4 0" "4 1"; do
  set -- $cfg
  export PLFEM_COLUMN_WGS_CAP=$1
  if [ "$2" = "-1" ]; then unset PLFEM_COLUMN_LEAN; else export PLFEM_COLUMN_LEAN=$2; fi
  rm -rf gpurun_out/prof_step
  rocprofv3 --kernel-trace -d gpurun_out/prof_step -o st --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_st.log 2>&1 || exit 1
  python3 scripts/factor_levels.py gpurun_out/prof_step/st_kernel_trace.csv > gpurun_out/levels_factor_cap$1_lean$2.txt
  echo "cap $1 lean $2"; grep -E "^total|L-[0-9]+ " gpurun_out/levels_factor_cap$1_lean$2.txt | cut -c1-18 | tr '\n' ' '; echo
done
rm -rf gpurun_out/prof_step
