#!/usr/bin/env python3
"""Time of the factorisation kernels from a rocprofv3 kernel trace (usage: factor_levels.py <kernel_trace.csv>):
microseconds and launches per factorisation, per kernel."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0])
nf = 0
for r in rows:
    m = re.search(r"(k_ldl_\w+(?:<\d>)?|k_front_gather|k_form_z|k_mirror_z|k_leaf_assemble)", r["Kernel_Name"])
    if not m:
        continue
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[m.group(1)][0] += d
    agg[m.group(1)][1] += 1
    nf += m.group(1) == "k_leaf_assemble"
print("factorisations in trace:", nf)
tot = 0.0
for name, (t, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{name:22s} {t / nf / 1e3:9.1f} us  {n // nf:5d} launches  avg {t / n / 1e3:7.2f} us")
    tot += t / nf / 1e3
print(f"total {tot:.1f} us per factorisation")
