"""Per-kernel-launch durations of scripts/sweep_filter_timing.py from the kernel-trace CSV: 3 phases x 5 solves."""
import csv, re, sys
import numpy as np
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
solves, cur = [], None
for r in rows:
    n = r["Kernel_Name"]
    if "k_permute_in" in n:
        cur = []
        solves.append(cur)
    elif cur is not None and re.search(r"k_(fwd|bwd)", n):
        cur.append((re.search(r"(k_\w+)", n).group(1), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
solves = [s for s in solves if len(s) == len(solves[0])]
per = len(solves) // 3
names = ["all fronts", "without s2 > 128", "only s2 > 128"]
tab = []
for ph in range(3):
    grp = solves[ph * per + 1:(ph + 1) * per]          # first solve of a phase = warm-up
    tab.append(np.mean([[k[2] for k in s] for s in grp], axis=0))
print(f"{'kernel':12s} {'wgs':>6s} " + " ".join(f"{n:>18s}" for n in names))
for i, k in enumerate(solves[0]):
    print(f"{k[0]:12s} {k[1]:6d} " + " ".join(f"{t[i]:18.2f}" for t in tab))
print(f"{'sum':12s} {'':6s} " + " ".join(f"{t.sum():18.1f}" for t in tab))
