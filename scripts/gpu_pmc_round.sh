#!/bin/bash
# The PMC passes of a round alone (HBM traffic per kernel family for the three ladder rungs, MFMA counters at C1):
#   gpu_pmc_round.sh <tag>  ->  gpurun_out/profiles_<tag>/<tag>_pmc_families_L{1,0,2}.json
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
for L in 1 0 2; do
  rm -rf gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma
  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof_fetch -o f --output-format csv -- python3 bench.py --levels $L --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_fetch_L$L.log 2>&1
  echo "fetch pass L=$L done"
  rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof_write -o w --output-format csv -- python3 bench.py --levels $L --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_write_L$L.log 2>&1
  echo "write pass L=$L done"
  MF=-
  if [ $L = 1 ]; then
    rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d gpurun_out/prof_mfma -o m --output-format csv -- python3 bench.py --levels $L --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_mfma.log 2>&1 || echo "mfma pass failed"
    [ -f gpurun_out/prof_mfma/m_counter_collection.csv ] && MF=gpurun_out/prof_mfma/m_counter_collection.csv
  fi
  python3 scripts/pmc_families.py gpurun_out/prof_fetch/f_counter_collection.csv gpurun_out/prof_write/w_counter_collection.csv $MF $OUT/${TAG}_pmc_families_L$L.json | tee $OUT/pmc_families_L$L.txt
done
rm -rf gpurun_out/prof_fetch gpurun_out/prof_write gpurun_out/prof_mfma
cp $OUT/${TAG}_pmc_families_L*.json profiles/
python3 bench.py --ladder --steps 5 --warmup 2 > $OUT/ladder.log 2>&1
grep '^{' $OUT/ladder.log > $OUT/${TAG}_ladder.json
python3 - <<PY
import json
for l in open("$OUT/${TAG}_ladder.json"):
    d = json.loads(l); r = d["roofline"]
    print("ladder", d["config"]["workload"][:28], round(d["ms_per_step"], 2), "ms  frac", round(r["frac"], 3), "traffic ratio", r["traffic_ratio"], r["pmc_check"], r["traffic_source"])
PY
