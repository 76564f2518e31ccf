#!/usr/bin/env python3
"""Timeline of one block Lanczos step from a rocprofv3 kernel trace (usage: step_timeline.py <kernel_trace.csv> [which]):
kernels between two consecutive k_block_scale launches, with start offsets, durations and the idle gap before each."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_block_scale" in r["Kernel_Name"]]
a, b = marks[which], marks[which + 1]
t0 = int(rows[a]["End_Timestamp"])
prev_end = t0
tot = gap_tot = 0.0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("plfem::(anonymous namespace)::", "").replace("void ", ""))
    g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{name:28s} wg={g:6d} start={(s - t0) / 1e3:8.1f} dur={(e - s) / 1e3:7.1f} gap={(s - prev_end) / 1e3:6.1f}")
    tot += (e - s) / 1e3
    gap_tot += (s - prev_end) / 1e3
    prev_end = e
print(f"step span {(prev_end - t0) / 1e3:.1f} us, kernels {tot:.1f} us, gaps {gap_tot:.1f} us, launches {b - a}")
