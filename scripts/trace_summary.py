#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV: per-level durations of one shift-invert solve and of one
factorisation (usage: trace_summary.py <kernel_trace.csv>)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def nm(s):
    m = re.search(r'(k_\w+|__amd_\w+)', s)
    return m.group(1) if m else s[:30]
names = [nm(r['Kernel_Name']) for r in rows]
dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows]
grid = [(int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z'])) for r in rows]
S = ('k_fwd_gather', 'k_fwd_own', 'k_fwd_bnd', 'k_bwd_t', 'k_bwd_x', 'k_fwd_own_dot', 'k_fwd_bnd_dot', 'k_bwd_t_dot', 'k_bwd_x_dot')
starts = [i for i in range(1, len(names)) if names[i] in S and names[i - 1] not in S]
if starts:
    s = starts[len(starts) // 2]
    e = s
    while e < len(names) and names[e] in S: e += 1
    t0 = int(rows[s]['Start_Timestamp'])
    for i in range(s, e):
        print(f"{names[i]:14s} grid={grid[i]} dur={dur[i]/1e3:7.1f}us start={(int(rows[i]['Start_Timestamp'])-t0)/1e3:8.1f}us")
    print('solve: launches', e - s, 'sum kernel us', sum(dur[s:e]) / 1e3, 'span us', (int(rows[e - 1]['End_Timestamp']) - t0) / 1e3)
FS = ('k_leaf_assemble', 'k_front_gather', 'k_ldl_diag', 'k_ldl_invrow', 'k_ldl_panel', 'k_ldl_update', 'k_form_z', 'k_mirror_z')
f0s = [i for i, n in enumerate(names) if n == 'k_leaf_assemble']
if f0s:
    f0 = f0s[-1]; f1 = f0
    while f1 < len(names) and names[f1] in FS: f1 += 1
    print('factor: launches', f1 - f0, 'span us', (int(rows[f1 - 1]['End_Timestamp']) - int(rows[f0]['Start_Timestamp'])) / 1e3, 'sum kernel us', sum(dur[f0:f1]) / 1e3)
    agg = {}
    for i in range(f0, f1): agg[names[i]] = agg.get(names[i], 0) + dur[i]
    print({k: round(v / 1e3, 1) for k, v in agg.items()})
    if len(sys.argv) > 2:
        t0 = int(rows[f0]['Start_Timestamp'])
        for i in range(f0, f1):
            print(f"{names[i]:14s} grid={grid[i]} dur={dur[i]/1e3:7.1f}us start={(int(rows[i]['Start_Timestamp'])-t0)/1e3:8.1f}us")
