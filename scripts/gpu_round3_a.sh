#!/bin/bash
# round 3 work loop: pivot-block micro benchmark, parity tests of the solve path, short bench
set -e -o pipefail
mkdir -p gpurun_out
./scripts/micro/build/permlane_check
./scripts/micro/build/pivot_bench | tail -1
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scalar.py -m gpu -x -q 2>&1 | tee gpurun_out/pytest_gpu_a.log | tail -15
PLFEM_LANCZOS_TRACE=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>gpurun_out/bench_tr.log >gpurun_out/bench_short.json
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_short.json") if l.startswith("{")][-1])
print(round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()},
      d["lanczos"], "frac", round(d["roofline"]["frac"], 3))
for k in d["roofline"]["kernels"]:
    print("  ", k["kernel"], round(k["achieved"], 1), k["unit"], "frac", round(k["frac"], 3), "avg_us", round(k["avg_us"], 1))
PY
tail -3 gpurun_out/bench_tr.log
./scripts/gpu_step_trace.sh | head -12
