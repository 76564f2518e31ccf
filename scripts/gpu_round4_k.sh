#!/bin/bash
# A/B on one box: kernel-argument preloading (Makefile PRELOAD) on / off -- parity tests, 20-step bench, per-level sweep table
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4k
mkdir -p $O
export PLFEM_MALLOC_TUNE=1
run_one() {
  tag=$1
  timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/bench_$tag.log >$O/bench_$tag.json
  python3 - $tag <<'PY'
import json, sys
tag = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/r4k/bench_{tag}.json") if l.startswith("{")][-1])
print(tag, round(d["value"], 1), "modes/s", round(d["ms_per_step"], 2), "ms", {k: round(v, 2) for k, v in d["breakdown_ms"].items()})
PY
  rm -rf gpurun_out/prof_stats
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_stats -o st --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/rocprof_$tag.log 2>&1
  cp gpurun_out/prof_stats/st_kernel_stats.csv $O/kernel_stats_$tag.csv
  python3 scripts/level_roofline.py gpurun_out/prof_stats/st_kernel_trace.csv > $O/levels_solve_$tag.txt
  python3 scripts/factor_levels.py gpurun_out/prof_stats/st_kernel_trace.csv > $O/levels_factor_$tag.txt
  rm -rf gpurun_out/prof_stats
  tail -2 $O/levels_solve_$tag.txt
  head -8 $O/levels_factor_$tag.txt
}
echo "== preload on (as built)"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_hfield_golden.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
run_one on
echo "== preload off (rebuilt here)"
make -C pl_fem_vectoriel_amd/csrc clean > /dev/null
make -C pl_fem_vectoriel_amd/csrc -j16 PRELOAD= > $O/make_off.log 2>&1 || { tail -30 $O/make_off.log; exit 1; }
run_one off
echo "== preload on again (rebuilt here: same box, later)"
make -C pl_fem_vectoriel_amd/csrc clean > /dev/null
make -C pl_fem_vectoriel_amd/csrc -j16 > $O/make_on.log 2>&1 || { tail -30 $O/make_on.log; exit 1; }
run_one on2
