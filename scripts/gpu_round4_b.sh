#!/bin/bash
# Round-4 work loop: parity tests of the solve path, host phases of the cold step, analysis timing, short bench.
set -e -o pipefail
mkdir -p gpurun_out/r4b
export PLFEM_MALLOC_TUNE=1
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scalar.py tests/test_hfield_golden.py tests/test_gpu_fullsize.py -m gpu -x -q > gpurun_out/r4b/pytest.log 2>&1 || { tail -40 gpurun_out/r4b/pytest.log; exit 1; }
tail -3 gpurun_out/r4b/pytest.log
timeout -k 10 300 python3 scripts/profile_python.py > gpurun_out/r4b/profile_python.txt 2>&1
tail -4 gpurun_out/r4b/profile_python.txt
for S in 0 4 6 8; do
  echo "== threads 16 side $S" >> gpurun_out/r4b/sym.txt
  PLFEM_SIDE_THREADS=$S PLFEM_HOST_THREADS=16 timeout -k 10 120 python3 scripts/time_symbolic.py 1 >> gpurun_out/r4b/sym.txt 2>&1
done
grep -v "^\[sym\]" gpurun_out/r4b/sym.txt
grep "^\[sym\]" gpurun_out/r4b/sym.txt | head -13
PLFEM_CTX_TRACE=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>gpurun_out/r4b/bench_ctx_trace.log >gpurun_out/r4b/bench_short.json
tail -2 gpurun_out/r4b/bench_ctx_trace.log
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4b/bench_short.json") if l.startswith("{")][-1])
print(round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()})
print(d["lanczos"], "frac", round(d["roofline"]["frac"], 3), d["step_ms"], d["host_ms_max"])
PY
