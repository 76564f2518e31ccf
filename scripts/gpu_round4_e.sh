#!/bin/bash
# panel-product micro-benchmark, then a short bench with the current defaults
set -e -o pipefail
mkdir -p gpurun_out/r4e
export PLFEM_MALLOC_TUNE=1
timeout -k 10 300 scripts/micro/build/panel_bench > gpurun_out/r4e/panel_bench.txt 2>&1
cat gpurun_out/r4e/panel_bench.txt
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/r4e/bench.log >gpurun_out/r4e/bench_short.json
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4e/bench_short.json") if l.startswith("{")][-1])
print(round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()})
print(d["lanczos"], "frac", round(d["roofline"]["frac"], 3), "pair us", round(d["roofline"]["avg_pair_us"], 1), d["step_ms"], d["host_ms_max"])
PY
