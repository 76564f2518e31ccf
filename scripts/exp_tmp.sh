cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -15 gpurun_out/pytest_gpu.log | cut -c1-200; exit 1; }; tail -2 gpurun_out/pytest_gpu.log
rm -rf gpurun_out/prof_lv
rocprofv3 --kernel-trace -d gpurun_out/prof_lv -o lv --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_lv.log 2>&1 || { tail -5 gpurun_out/bench_lv.log; exit 1; }
python3 scripts/factor_levels.py gpurun_out/prof_lv/lv_kernel_trace.csv | cut -c1-160
rm -rf gpurun_out/prof_lv
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | grep "^{" | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); b=d['breakdown_ms']; print(round(d['value'],1), round(d['ms_per_step'],2), {k: round(v,2) for k,v in b.items()})"
