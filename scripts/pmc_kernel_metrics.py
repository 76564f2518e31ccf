#!/usr/bin/env python3
"""Average of every collected counter per kernel (template arguments kept, signature dropped) from rocprofv3
--pmc counter_collection CSVs: pmc_kernel_metrics.py <kernel-regex> <csv> [<csv> ...]"""
import collections, csv, re, sys
pat = re.compile(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[2:]:
    per = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        m = re.search(r"(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if not m or not pat.search(m.group(1)):
            continue
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per[key] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = (m.group(1), r.get("Grid_Size", ""))
    for (disp, cname), v in per.items():
        a = acc[names[disp]][cname]
        a[0] += v
        a[1] += 1
for k in sorted(acc):
    print(f"{k[0]:24s} grid={k[1]:>9s} " + "  ".join(f"{c}={v[0] / v[1]:.3g}" for c, v in sorted(acc[k].items())))
