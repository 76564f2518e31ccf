#!/bin/bash
# Host analysis timing on the GPU box's host (no GPU work): thread counts, side-team share, subtree tasks.
set -e -o pipefail
mkdir -p gpurun_out/r4sym
export PLFEM_MALLOC_TUNE=1
O=gpurun_out/r4sym/sym.txt
: > $O
for T in 8 16 24; do
  echo "== threads $T" >> $O
  PLFEM_HOST_THREADS=$T timeout -k 10 120 python3 scripts/time_symbolic.py 1 >> $O 2>&1
done
for SIDE in 2 3 6; do
  echo "== threads 16 side $SIDE" >> $O
  PLFEM_SIDE_THREADS=$SIDE PLFEM_HOST_THREADS=16 timeout -k 10 120 python3 scripts/time_symbolic.py 1 2>&1 | grep -v "^\[sym\]" >> $O
done
for SUB in 16 64; do
  echo "== threads 16 subtrees $SUB" >> $O
  PLFEM_TREE_SUBTREES=$SUB PLFEM_HOST_THREADS=16 timeout -k 10 120 python3 scripts/time_symbolic.py 1 2>&1 | grep -v "^\[sym\]" >> $O
done
echo "== L=2, threads 16" >> $O
PLFEM_HOST_THREADS=16 timeout -k 10 120 python3 scripts/time_symbolic.py 2 >> $O 2>&1
cat $O
