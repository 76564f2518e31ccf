#!/bin/bash
# the whole timed -m gpu suite, smoke(), then the perf-only loop
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r4g/pytest.log 2>&1 || { tail -60 gpurun_out/r4g/pytest.log; exit 1; }
tail -14 gpurun_out/r4g/pytest.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_round4_f.sh
