#!/bin/bash
# quick loop: a few parity tests, kernel stats of a short bench, then the 20-step bench
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
export PLFEM_MALLOC_TUNE=1
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_hfield_golden.py -m gpu -x -q > gpurun_out/r4h/pytest.log 2>&1 || { tail -40 gpurun_out/r4h/pytest.log; exit 1; }
tail -2 gpurun_out/r4h/pytest.log
rm -rf gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o st --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r4h/bench_under_rocprof.log 2>&1
cp gpurun_out/prof_stats/st_kernel_stats.csv gpurun_out/r4h/kernel_stats.csv
python3 scripts/level_roofline.py gpurun_out/prof_stats/st_kernel_trace.csv > gpurun_out/r4h/levels_solve.txt
python3 scripts/factor_levels.py gpurun_out/prof_stats/st_kernel_trace.csv > gpurun_out/r4h/levels_factor.txt
rm -rf gpurun_out/prof_stats
sed 's/plfem::(anonymous namespace):://g' gpurun_out/r4h/kernel_stats.csv | cut -d'(' -f1,2 | cut -c1-70 > /dev/null
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r4h/kernel_stats.csv")))
for r in rows[:34]:
    n = r["Name"].replace("plfem::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print(f"{n:34s} calls {int(r['Calls']):5d}  total {float(r['TotalDurationNs'])/8e3:9.1f} us/solve  avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
head -9 gpurun_out/r4h/levels_factor.txt
tail -2 gpurun_out/r4h/levels_solve.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>gpurun_out/r4h/bench.log >gpurun_out/r4h/bench_short.json
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4h/bench_short.json") if l.startswith("{")][-1])
print(round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()})
print(d["step_ms"], d["host_ms_max"])
PY
