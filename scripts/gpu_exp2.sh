#!/bin/bash
# GPU tests, short bench, per-level sweep / per-kernel factorisation tables from a kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -q ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_short.json 2>gpurun_out/bench_tr.log || { tail -20 gpurun_out/bench_tr.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_short.json") if l.startswith("{")][-1])
print(round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()},
      d["lanczos"], "frac", round(d["roofline"]["frac"], 3))
for k in d["roofline"]["kernels"]:
    print("  ", k["kernel"], round(k["achieved"], 1), k["unit"], "frac", round(k["frac"], 3), "avg_us", round(k["avg_us"], 1))
PY
rm -rf gpurun_out/prof_lv
rocprofv3 --kernel-trace -d gpurun_out/prof_lv -o lv --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_lv.log 2>&1 || exit 1
python3 scripts/level_roofline.py gpurun_out/prof_lv/lv_kernel_trace.csv > gpurun_out/levels_solve.txt
python3 scripts/factor_levels.py gpurun_out/prof_lv/lv_kernel_trace.csv > gpurun_out/levels_factor.txt
cat gpurun_out/levels_solve.txt; head -14 gpurun_out/levels_factor.txt
rm -f gpurun_out/prof_lv/lv_kernel_trace.csv
