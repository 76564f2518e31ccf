#!/bin/bash
# perf-only loop: host phases of the cold step, analysis timing, a 20-step bench
set -e -o pipefail
mkdir -p gpurun_out/r4f
export PLFEM_MALLOC_TUNE=1
timeout -k 10 300 python3 scripts/profile_python.py > gpurun_out/r4f/profile_python.txt 2>&1
tail -4 gpurun_out/r4f/profile_python.txt
PLFEM_HOST_THREADS=16 timeout -k 10 120 python3 scripts/time_symbolic.py 1 > gpurun_out/r4f/sym.txt 2>&1
grep -v "^\[sym\]" gpurun_out/r4f/sym.txt
grep "^\[sym\]" gpurun_out/r4f/sym.txt | tail -13
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4f/bench.log >gpurun_out/r4f/bench_short.json
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4f/bench_short.json") if l.startswith("{")][-1])
print(round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()})
print(d["lanczos"], "frac", round(d["roofline"]["frac"], 3), "pair us", round(d["roofline"]["avg_pair_us"], 1), d["step_ms"], d["host_ms_max"])
PY
