#!/bin/bash
# parity tests of the solve path + level table of the sweeps (rocprofv3 kernel trace of a short bench run)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scalar.py -x -q 2>&1 | tail -3
rm -rf gpurun_out/prof_lv
rocprofv3 --kernel-trace -d gpurun_out/prof_lv -o lv --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_lv.log 2>&1
grep '^{' gpurun_out/bench_lv.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', {k: round(v,2) for k,v in d['breakdown_ms'].items()})"
python3 scripts/level_roofline.py gpurun_out/prof_lv/lv_kernel_trace.csv | grep -E "k_bwd|k_fwd\*"
rm -rf gpurun_out/prof_lv
python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2), 'ms', {k: round(v,2) for k,v in d['breakdown_ms'].items()})"
