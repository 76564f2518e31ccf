#!/bin/bash
# Round-4 work loop: the whole -m gpu suite (timed), analysis timing, short bench, kernel stats of a short bench run.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
export PLFEM_MALLOC_TUNE=1
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r4c/pytest.log 2>&1 || { tail -60 gpurun_out/r4c/pytest.log; exit 1; }
tail -18 gpurun_out/r4c/pytest.log
for S in 0 3 8; do
  echo "== threads 16 side $S" >> gpurun_out/r4c/sym.txt
  PLFEM_SIDE_THREADS=$S PLFEM_HOST_THREADS=16 timeout -k 10 120 python3 scripts/time_symbolic.py 1 >> gpurun_out/r4c/sym.txt 2>&1
done
grep -v "^\[sym\]" gpurun_out/r4c/sym.txt
grep "^\[sym\]" gpurun_out/r4c/sym.txt | head -13
PLFEM_CTX_TRACE=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/r4c/bench_ctx_trace.log >gpurun_out/r4c/bench_short.json
tail -2 gpurun_out/r4c/bench_ctx_trace.log
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4c/bench_short.json") if l.startswith("{")][-1])
print(round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()})
print(d["lanczos"], "frac", round(d["roofline"]["frac"], 3), d["step_ms"], d["host_ms_max"])
PY
rm -rf gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o st --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4c/bench_under_rocprof.log 2>&1
cp gpurun_out/prof_stats/st_kernel_stats.csv gpurun_out/r4c/kernel_stats.csv
python3 scripts/level_roofline.py gpurun_out/prof_stats/st_kernel_trace.csv > gpurun_out/r4c/levels_solve.txt
python3 scripts/factor_levels.py gpurun_out/prof_stats/st_kernel_trace.csv > gpurun_out/r4c/levels_factor.txt
rm -rf gpurun_out/prof_stats
head -24 gpurun_out/r4c/kernel_stats.csv | cut -c1-170
