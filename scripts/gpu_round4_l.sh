#!/bin/bash
# modes per post-processing group (PLFEM_POST_GROUP): 20-step bench each, the phases that move
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4l
mkdir -p $O
export PLFEM_MALLOC_TUNE=1
for G in 8 22 11 4 8 22; do
  PLFEM_POST_GROUP=$G timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_$G.log >$O/bench_$G.json
  python3 - $G <<'PY'
import json, sys
G = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/r4l/bench_{G}.json") if l.startswith("{")][-1])
b = d["breakdown_ms"]
print("group", G, round(d["ms_per_step"], 2), "ms | post", round(b["post"], 3), "resid", round(b["residual_check"], 3), "gaps", round(b["call_gaps"], 3),
      "post+resid+gaps", round(b["post"] + b["residual_check"] + b["call_gaps"], 3), "| symbolic", round(b["symbolic_host"], 2), "lanczos", round(b["lanczos"], 2))
PY
done
