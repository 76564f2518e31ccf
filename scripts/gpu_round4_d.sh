#!/bin/bash
# Sweep kernels with 16-byte loads: parity tests, then kernel stats of a short bench for the loads-in-flight variants.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
export PLFEM_MALLOC_TUNE=1
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scalar.py tests/test_hfield_golden.py -m gpu -x -q > gpurun_out/r4d/pytest.log 2>&1 || { tail -40 gpurun_out/r4d/pytest.log; exit 1; }
tail -3 gpurun_out/r4d/pytest.log
PLFEM_HOST_THREADS=16 timeout -k 10 120 python3 scripts/time_symbolic.py 1 > gpurun_out/r4d/sym.txt 2>&1
grep -v "^\[sym\]" gpurun_out/r4d/sym.txt
for V in "8 8" "4 4" "8 4" "4 8"; do
  set -- $V
  export PLFEM_SWEEP_LD_TILE=$1 PLFEM_SWEEP_LD_ROWS=$2
  rm -rf gpurun_out/prof_stats
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o st --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4d/bench_$1_$2.log 2>&1
  cp gpurun_out/prof_stats/st_kernel_stats.csv gpurun_out/r4d/kernel_stats_$1_$2.csv
  python3 scripts/level_roofline.py gpurun_out/prof_stats/st_kernel_trace.csv > gpurun_out/r4d/levels_solve_$1_$2.txt
  echo "== tile $1 rows $2"
  grep -E "k_fwd|k_bwd" gpurun_out/r4d/kernel_stats_$1_$2.csv | sed 's/plfem::(anonymous namespace):://g; s/(SweepArgs)//' | cut -d, -f1-4
  python3 - <<PY
import json
d = json.loads([l for l in open("gpurun_out/r4d/bench_$1_$2.log") if l.startswith("{")][-1])
print(round(d["ms_per_step"], 2), "ms  lanczos", round(d["breakdown_ms"]["lanczos"], 2), "pair us", round(d["roofline"]["avg_pair_us"], 1))
PY
done
tail -20 gpurun_out/r4d/levels_solve_8_8.txt
