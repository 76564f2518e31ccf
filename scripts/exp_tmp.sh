cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in 32 512; do
  export PLFEM_FWD_ROWS_MAX=$cfg
  rm -rf gpurun_out/prof_lv
  rocprofv3 --kernel-trace -d gpurun_out/prof_lv -o lv --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_lv.log 2>&1 || { tail -5 gpurun_out/bench_lv.log; exit 1; }
  echo "== FWD_ROWS_MAX=$cfg"; python3 scripts/level_roofline.py gpurun_out/prof_lv/lv_kernel_trace.csv | grep "k_fwd" | cut -c1-130
done
