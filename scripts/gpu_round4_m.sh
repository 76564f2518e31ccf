#!/bin/bash
# A/B on one box: spin-then-block waits where the host is on the critical path (PLFEM_SPIN_US = 0: block at once)
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m
mkdir -p $O
export PLFEM_MALLOC_TUNE=1
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -1 $O/pytest.log
for S in 0 1500 0 1500 0 1500 300; do
  PLFEM_SPIN_US=$S timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_$S.log >$O/bench_$S.json
  python3 - $S <<'PY'
import json, sys
S = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/r4m/bench_{S}.json") if l.startswith("{")][-1])
b = d["breakdown_ms"]
ms = sorted(d["step_ms"])
print("spin us", S, "mean", round(d["ms_per_step"], 2), "median", ms[len(ms) // 2], "| lanczos", round(b["lanczos"], 3), "gaps", round(b["call_gaps"], 3), "python", round(b["python"], 2),
      "symbolic", round(b["symbolic_host"], 2), "| warm", round(b["warm_step"], 2))
PY
done
