#!/bin/bash
# Collects the committed profile artefacts of a round on the GPU box (run through gpurun):
#   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`
#   profiles/<tag>_pmc_k_fwd.json     FETCH_SIZE / WRITE_SIZE of the dominant kernel (separate --pmc passes)
#   profiles/<tag>_bench_line.json    the bench line of a plain run (with the CPU baseline)
#   profiles/<tag>_levels_*.txt       per-level sweep table / per-level factorisation table from a kernel trace
#   profiles/<tag>_ladder.json        bench.py --ladder (BASELINE configs[2]: L = 0, 1, 2)
#   profiles/<tag>_sweep_1gpu.json    bench.py --sweep on one GPU (BASELINE configs[3])
# usage: gpu_profile_round.sh <tag>      (outputs land in gpurun_out/profiles_<tag>/ for copying into profiles/)
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$TAG
rm -rf $OUT gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_lv && mkdir -p $OUT
python3 bench.py > $OUT/bench_full.log 2>&1
grep '^{' $OUT/bench_full.log > $OUT/${TAG}_bench_line.json
echo "bench line done"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o st --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
cp gpurun_out/prof_stats/st_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 scripts/level_roofline.py gpurun_out/prof_stats/st_kernel_trace.csv > $OUT/${TAG}_levels_solve.txt
python3 scripts/factor_levels.py gpurun_out/prof_stats/st_kernel_trace.csv > $OUT/${TAG}_levels_factor.txt
rm -rf gpurun_out/prof_stats
echo "kernel stats done"
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
python3 scripts/pmc_summary.py 'k_fwd(_mix)?<4' gpurun_out/prof_fetch/f_counter_collection.csv gpurun_out/prof_write/w_counter_collection.csv $OUT/${TAG}_pmc_k_fwd.json
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write
echo "pmc done"
python3 bench.py --ladder --steps 5 --warmup 2 > $OUT/ladder.log 2>&1
grep '^{' $OUT/ladder.log > $OUT/${TAG}_ladder.json
python3 bench.py --sweep --steps 2 --warmup 1 > $OUT/sweep.log 2>&1
grep '^{' $OUT/sweep.log > $OUT/${TAG}_sweep_1gpu.json
head -12 $OUT/${TAG}_kernel_stats.csv | cut -c1-150
cut -c1-400 $OUT/${TAG}_bench_line.json
python3 - <<PY
import json
for l in open("$OUT/${TAG}_ladder.json"):
    d = json.loads(l); print("ladder", d["config"]["workload"][:40], round(d["ms_per_step"], 2), "ms", round(d["roofline"]["frac"], 3))
d = json.loads(open("$OUT/${TAG}_sweep_1gpu.json").read()); print("sweep", round(d["ms_per_step"], 1), "ms", round(d["value"], 1), "modes/s")
PY
