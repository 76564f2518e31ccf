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
    m = re.search(r"(k_ldl_\w+(?:<\d>)?|k_front_gather|k_form_z|k_mirror_z|k_mirror_x|k_leaf_assemble)", r["Kernel_Name"])
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

# per tree level of the LAST factorisation in the trace: a level starts at k_leaf_assemble / k_front_gather
seq = []
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    m = re.search(r"(k_ldl_\w+(?:<\d>)?|k_front_gather|k_form_z|k_mirror_z|k_mirror_x|k_leaf_assemble)", r["Kernel_Name"])
    if m:
        seq.append((m.group(1), int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
last = max(i for i, s in enumerate(seq) if s[0] == "k_leaf_assemble")
seq = seq[last:]
levels, cur = [], None
for name, t0, t1, wg in seq:
    if name in ("k_leaf_assemble", "k_front_gather"):
        cur = {"t0": t0, "t1": t1, "k": collections.defaultdict(lambda: [0, 0])}
        levels.append(cur)
    cur["t1"] = t1
    cur["k"][name][0] += t1 - t0
    cur["k"][name][1] += 1
print("level (from the leaves)  wall us | per kernel: busy us (launches)")
for i, lv in enumerate(levels):
    parts = "  ".join(f"{n.replace('k_ldl_', '').replace('k_', '')} {v[0] / 1e3:.0f}({v[1]})" for n, v in lv["k"].items())
    print(f"  L-{i:<2d} {(lv['t1'] - lv['t0']) / 1e3:8.1f} | {parts}")
