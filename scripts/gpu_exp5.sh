#!/bin/bash
# bench + host analysis timing (no tests)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PLFEM_SYM_TRACE= python3 scripts/time_symbolic.py 8 16 2>&1 | tail -2
PLFEM_SYM_TRACE=1 python3 scripts/time_symbolic.py 8 2>&1 | grep "^\[sym\]" | tail -13
timeout -k 10 300 python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_short.json 2>gpurun_out/bench_tr.log || { tail -20 gpurun_out/bench_tr.log; exit 1; }
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/bench_short.json") if l.startswith("{")][-1])
print(round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()},
      d["lanczos"], "frac", round(d["roofline"]["frac"], 3), d["step_ms"])
PY
