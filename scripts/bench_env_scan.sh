#!/bin/bash
# headline bench for several values of one environment variable: bench_env_scan.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['ms_per_step'],2), 'ms', {k: round(x,2) for k,x in d['breakdown_ms'].items()}, 'OP', d['lanczos']['n_opinv'])"
done
