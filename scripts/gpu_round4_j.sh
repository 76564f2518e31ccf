#!/bin/bash
# VERDICT r3 item 4, second half: the block width of the Lanczos recurrence.  P = 4 (the product) against a library built
# with -DPLFEM_BLOCK_P=8 ON THE BOX (the box's copy of the tree is thrown away afterwards): basis-size scan of the Lanczos
# run alone, parity tests of the solve path, 20-step bench, kernel stats + per-level sweep table.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j
mkdir -p $O
export PLFEM_MALLOC_TUNE=1
echo "== P = 4"
NCVS=104,120,132,144,160 TOL=1e-8 timeout -k 10 300 python3 scripts/ncv_sweep.py 1 22 2>&1 | tee $O/ncv_p4.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_p4.log >$O/bench_p4.json
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4j/bench_p4.json") if l.startswith("{")][-1])
print("P=4", round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()}, d["lanczos"])
PY
echo "== P = 8 (rebuilt here)"
make -C pl_fem_vectoriel_amd/csrc clean > /dev/null
make -C pl_fem_vectoriel_amd/csrc -j16 EXTRA_CXXFLAGS=-DPLFEM_BLOCK_P=8 > $O/make_p8.log 2>&1 || { tail -30 $O/make_p8.log; exit 1; }
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "eigenpairs or mode_records or residuals or north_star or shift_invert or every_arrangement" > $O/pytest_p8.log 2>&1 || { tail -40 $O/pytest_p8.log; exit 1; }
tail -2 $O/pytest_p8.log
NCVS=96,104,120,136,144,160 TOL=1e-8 timeout -k 10 300 python3 scripts/ncv_sweep.py 1 22 2>&1 | tee $O/ncv_p8.txt
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_p8.log >$O/bench_p8.json
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r4j/bench_p8.json") if l.startswith("{")][-1])
print("P=8", round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()}, d["lanczos"])
PY
rm -rf gpurun_out/prof_stats
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o st --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
cp gpurun_out/prof_stats/st_kernel_stats.csv $O/kernel_stats_p8.csv
python3 scripts/level_roofline.py gpurun_out/prof_stats/st_kernel_trace.csv > $O/levels_solve_p8.txt
rm -rf gpurun_out/prof_stats
python3 - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/r4j/kernel_stats_p8.csv")))
for r in rows[:26]:
    n = r["Name"].replace("plfem::(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    print(f"{n:34s} calls {int(r['Calls']):5d}  total {float(r['TotalDurationNs'])/4e3:9.1f} us/solve  avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
cat $O/levels_solve_p8.txt | tail -30
