#!/bin/bash
# Collects the committed profile artefacts of a round on the GPU box (run through gpurun):
#   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `bench.py --steps 3 --warmup 1`
#   profiles/<tag>_pmc_k_fwd.json     FETCH_SIZE / WRITE_SIZE of the dominant kernel (separate --pmc passes)
#   profiles/<tag>_bench_line.json    the bench line of a plain run
# usage: gpu_profile_round.sh <tag>      (outputs land in gpurun_out/profiles_<tag>/ for copying into profiles/)
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$TAG
rm -rf $OUT gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write && mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o st --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
cp gpurun_out/prof_stats/st_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o f --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -o w --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_write.log 2>&1
python3 scripts/pmc_summary.py 'k_fwd<4' gpurun_out/prof_fetch/f_counter_collection.csv gpurun_out/prof_write/w_counter_collection.csv $OUT/${TAG}_pmc_k_fwd.json
python3 bench.py > $OUT/bench_full.log 2>&1
grep '^{' $OUT/bench_full.log > $OUT/${TAG}_bench_line.json
head -12 $OUT/${TAG}_kernel_stats.csv | cut -c1-150
cut -c1-400 $OUT/${TAG}_bench_line.json
