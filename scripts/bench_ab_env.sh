#!/bin/bash
# A/B of one environment switch on the headline bench, interleaved: bench_ab_env.sh VAR=VALUE [rounds]
set -e
SW=$1; N=${2:-3}
for i in $(seq $N); do
  python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('default      ', round(d['ms_per_step'],2), 'ms', {k: round(v,2) for k,v in d['breakdown_ms'].items()})"
  env $SW python3 bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$SW', round(d['ms_per_step'],2), 'ms', {k: round(v,2) for k,v in d['breakdown_ms'].items()})"
done
