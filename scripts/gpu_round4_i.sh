#!/bin/bash
# scans: analysis thread counts, leaf size of the front tree, then the 20-step bench at the defaults
set -e -o pipefail
mkdir -p gpurun_out/r4i
export PLFEM_MALLOC_TUNE=1
O=gpurun_out/r4i/scan.txt
: > $O
for T in 12 16 20 24 32; do
  echo "== analysis threads $T" >> $O
  PLFEM_HOST_THREADS=$T timeout -k 10 120 python3 scripts/time_symbolic.py 1 2>&1 | grep -v "^\[sym\]" >> $O
done
for LEAF in 24 48; do
  echo "== leaf $LEAF" >> $O
  PLFEM_LEAF_ELEMS=$LEAF timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['ms_per_step'],2),'ms',{k:round(v,2) for k,v in d['breakdown_ms'].items() if k in ('symbolic_host','context','factor','lanczos','call_gaps','python','warm_step')}, 'pair us', round(d['roofline']['avg_pair_us'],1))" >> $O
done
for T in 12 20 24; do
  echo "== bench with analysis threads $T" >> $O
  PLFEM_HOST_THREADS=$T timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(round(d['ms_per_step'],2),'ms',{k:round(v,2) for k,v in d['breakdown_ms'].items() if k in ('symbolic_host','context','python')}, d['host_ms_max'])" >> $O
done
cat $O
bash scripts/gpu_round4_h.sh
