#!/bin/bash
# full GPU suite, smoke, default bench, 2-rank rehearsals on one GPU (gloo, both ranks on device 0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -m gpu -q 2>&1 | tee gpurun_out/pytest_gpu.log | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py > gpurun_out/bench_full.log 2>&1; grep '^{' gpurun_out/bench_full.log > gpurun_out/bench_line_final.json; cut -c1-330 gpurun_out/bench_line_final.json
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_line_final.json").read())
r = d["roofline"]
print("roofline", round(r["frac"], 3), round(r["avg_pair_us"], 1), "us per pair; traffic ratio", r["traffic_ratio"], "| step_ms", d["step_ms"])
PY
PLFEM_BENCH_BACKEND=gloo PLFEM_BENCH_SAME_DEVICE=1 python3 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline 2>gpurun_out/two_rank.log | cut -c1-300
PLFEM_BENCH_BACKEND=gloo PLFEM_BENCH_SAME_DEVICE=1 python3 bench.py --gpus 2 --sweep --steps 1 --warmup 0 2>>gpurun_out/two_rank.log | cut -c1-300
tail -3 gpurun_out/two_rank.log
