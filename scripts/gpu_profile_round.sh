#!/bin/bash
# Collects the committed profile artefacts of a round on the GPU box (run through gpurun):
#   profiles/<tag>_bench_line.json     the bench line of a plain run (with the CPU baseline)
#   profiles/<tag>_kernel_stats.csv    rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`
#   profiles/<tag>_levels_*.txt        per-level sweep table / per-level factorisation table from that kernel trace
#   profiles/<tag>_pmc_families_L<L>.json   FETCH_SIZE / WRITE_SIZE per kernel family (separate --pmc passes) + MFMA counters,
#                                      one file per ladder rung L = 1 (C1), 0, 2: bench.py reads the file of ITS workload
#   profiles/<tag>_ladder.json         bench.py --ladder (BASELINE configs[2]: L = 0, 1, 2)
#   profiles/<tag>_sweep_1gpu.json     bench.py --sweep on one GPU (BASELINE configs[3]), 1 / 2 / 4 lanes
#   profiles/<tag>_lane_overlap.txt    kernel-trace summary of the 4-lane sweep (how many kernels overlap, per HW queue)
# usage: gpu_profile_round.sh <tag>      (outputs land in gpurun_out/profiles_<tag>/ for copying into profiles/)
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$TAG
rm -rf $OUT gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma gpurun_out/prof_lanes && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o st --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
cp gpurun_out/prof_stats/st_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python3 scripts/level_roofline.py gpurun_out/prof_stats/st_kernel_trace.csv > $OUT/${TAG}_levels_solve.txt
python3 scripts/factor_levels.py gpurun_out/prof_stats/st_kernel_trace.csv > $OUT/${TAG}_levels_factor.txt
rm -rf gpurun_out/prof_stats
echo "kernel stats done"
bash scripts/gpu_pmc_round.sh $TAG > $OUT/pmc_round.log 2>&1      # (PMC passes per ladder rung + the ladder lines that read them)
tail -4 $OUT/pmc_round.log
cp $OUT/${TAG}_pmc_families_L*.json profiles/     # (on the box: the bench lines below read the traffic of THIS round's passes)
echo "pmc done"
python3 bench.py --cpu-all-cores > $OUT/bench_full.log 2>&1
grep '^{' $OUT/bench_full.log > $OUT/${TAG}_bench_line.json
echo "bench line done"
: > $OUT/${TAG}_sweep_1gpu.json
for L in 1 2 4; do
  python3 bench.py --sweep --steps 2 --warmup 1 --lanes $L > $OUT/sweep_$L.log 2>&1
  grep '^{' $OUT/sweep_$L.log >> $OUT/${TAG}_sweep_1gpu.json
  echo "sweep lanes $L done"
done
rocprofv3 --kernel-trace -d gpurun_out/prof_lanes -o ln --output-format csv -- python3 bench.py --sweep --steps 1 --warmup 0 --lanes 4 > $OUT/sweep_trace.log 2>&1
python3 scripts/lane_overlap.py gpurun_out/prof_lanes/ln_kernel_trace.csv > $OUT/${TAG}_lane_overlap.txt
rm -rf gpurun_out/prof_lanes
cat $OUT/${TAG}_lane_overlap.txt
head -12 $OUT/${TAG}_kernel_stats.csv | cut -c1-150
cut -c1-600 $OUT/${TAG}_bench_line.json
python3 - <<PY
import json
for l in open("$OUT/${TAG}_ladder.json"):
    d = json.loads(l); print("ladder", d["config"]["workload"][:40], round(d["ms_per_step"], 2), "ms", round(d["roofline"]["frac"], 3))
for l in open("$OUT/${TAG}_sweep_1gpu.json"):
    d = json.loads(l); print("sweep", d["config"]["parallelism"], round(d["ms_per_step"], 1), "ms", round(d["value"], 1), "modes/s", "with mesh production", round(d["sweep"]["with_mesh_production"]["ms_per_step"], 1), "ms")
PY
