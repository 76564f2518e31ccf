#!/bin/bash
# round-2 experiment 1: full GPU tests, bench, per-level traces, host analysis by thread count, Lanczos tolerance
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_short.json 2>gpurun_out/bench_tr.log || { tail -20 gpurun_out/bench_tr.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_short.json") if l.startswith("{")][-1])
print(round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()},
      d["lanczos"], "frac", round(d["roofline"]["frac"], 3))
for k in d["roofline"]["kernels"]:
    print("  ", k["kernel"], round(k["achieved"], 1), k["unit"], "frac", round(k["frac"], 3), "avg_us", round(k["avg_us"], 1))
PY
echo "--- host analysis by threads"
PLFEM_SYM_TRACE= python3 scripts/time_symbolic.py 4 8 12 16 2>&1 | tail -4
PLFEM_SYM_TRACE=1 python3 scripts/time_symbolic.py 8 2>&1 | tail -40 > gpurun_out/sym_trace.txt
echo "--- per-level traces"
rm -rf gpurun_out/prof_lv
rocprofv3 --kernel-trace -d gpurun_out/prof_lv -o lv --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_lv.log 2>&1 || exit 1
python3 scripts/level_roofline.py gpurun_out/prof_lv/lv_kernel_trace.csv > gpurun_out/levels_solve.txt
python3 scripts/factor_levels.py gpurun_out/prof_lv/lv_kernel_trace.csv > gpurun_out/levels_factor.txt
grep sweep gpurun_out/levels_solve.txt; head -12 gpurun_out/levels_factor.txt
