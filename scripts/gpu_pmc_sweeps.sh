#!/bin/bash
# SQ counters of the sweep kernels (two --pmc passes, no other tracing)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -d gpurun_out/pmc1 -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc1.log 2>&1 || { tail -5 gpurun_out/pmc1.log; exit 1; }
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -d gpurun_out/pmc2 -o p --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc2.log 2>&1 || { tail -5 gpurun_out/pmc2.log; exit 1; }
python3 scripts/pmc_kernel_metrics.py 'k_fwd|k_bwd' gpurun_out/pmc1/p_counter_collection.csv > gpurun_out/pmc_sweeps1.txt
python3 scripts/pmc_kernel_metrics.py 'k_fwd|k_bwd' gpurun_out/pmc2/p_counter_collection.csv > gpurun_out/pmc_sweeps2.txt
ls gpurun_out/pmc1 | head; cut -c1-400 gpurun_out/pmc_sweeps1.txt | head -40
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
cut -c1-400 gpurun_out/pmc_sweeps2.txt | head -40
echo "--- scalar + quick GPU tests"
timeout -k 10 600 python3 -m pytest tests/test_gpu_scalar.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; tail -15 gpurun_out/pytest_gpu.log
