#!/usr/bin/env python3
"""Per-level time of the factorisation kernels from a rocprofv3 kernel trace (usage: factor_levels.py <kernel_trace.csv>).
Columns: total microseconds / launches per factorisation, one row per tree level (identified by its front count)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0])
nf = 0
for r in rows:
    m = re.search(r"(k_ldl_\w+|k_front_gather|k_form_z|k_mirror_z|k_leaf_assemble)", r["Kernel_Name"])
    if not m:
        continue
    name = m.group(1)
    wg = [int(r["Workgroup_Size_" + a]) for a in "XYZ"]
    g = [int(r["Grid_Size_" + a]) // w for a, w in zip("XYZ", wg)]
    cnt = g[0] if name in ("k_ldl_diag", "k_leaf_assemble") else (g[1] if name in ("k_ldl_invrow", "k_ldl_panel", "k_ldl_invrow_panel") else g[2])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[(cnt, name)][0] += d
    agg[(cnt, name)][1] += 1
    nf += name == "k_leaf_assemble"
names = ["k_leaf_assemble", "k_front_gather", "k_ldl_diag", "k_ldl_invrow_panel", "k_ldl_update", "k_form_z", "k_mirror_z"]
print("factorisations in trace:", nf)
print("fronts  " + " ".join(f"{n[2:]:>16s}" for n in names) + "   total_us")
T = 0.0
for cnt in sorted({k[0] for k in agg}):
    line, tot = f"{cnt:6d}  ", 0.0
    for n in names:
        v = agg.get((cnt, n), [0, 0])
        line += f"{v[0] / nf / 1e3:11.1f}/{v[1] // nf:<4d} "
        tot += v[0] / nf / 1e3
    print(line, f"{tot:8.1f}")
    T += tot
print(f"total {T:.1f} us per factorisation")
