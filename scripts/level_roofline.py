#!/usr/bin/env python3
"""Per-level roofline table of the triangular-solve kernels from a rocprofv3 kernel trace of bench.py
(usage: level_roofline.py <kernel_trace.csv> [levels]).  Bytes per level are the algorithmic ones of
DESIGN.md (the triangle of [L11^-1 | Z] resp. [L11^-T | Z^T] once, plus P right-hand sides in and out)."""
import csv
import json
import re
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh, _native  # noqa: E402

levels = int(sys.argv[2]) if len(sys.argv) > 2 else 1
geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(geom, refinement=1.0, levels=levels)
sym = _native.Symbolic(mesh.p, mesh.t)
fs = 2 * sym.array("fs").astype(np.int64)
fb = 2 * sym.array("fb").astype(np.int64)
nf = len(fs)
lev = np.floor(np.log2(np.arange(nf) + 1)).astype(int)
nlev = lev.max() + 1
mat = np.zeros(nlev)
vec = np.zeros(nlev)
cnt = np.zeros(nlev, dtype=int)
for L in range(nlev):
    s, b = fs[lev == L], fb[lev == L]
    m = s + b
    mat[L] = 8.0 * (s * m - 0.5 * s * s).sum()
    vec[L] = 8.0 * (m + s).sum()
    cnt[L] = (lev == L).sum()

# workgroups per level of each sweep (the compact launch lists of the context; rules of csrc/device.h + create_impl:
# pure row form up to 32 fronts, otherwise a mixed launch -- tiles of 64 rows, row-form workgroups of 16 rows for the
# fronts with more than 192 owned DOFs in the forward sweep)
def fwd_rows(count): return 8 if count <= 8 else 16 if count <= 32 else 64
def bwd_rows(count, leaf): return 64 if leaf else 8 if count <= 32 else 16
nblk = {}
for L in range(nlev):
    s_, m_ = fs[lev == L], fs[lev == L] + fb[lev == L]
    leaf = L == nlev - 1
    fr, br = fwd_rows(cnt[L]), bwd_rows(cnt[L], leaf)
    big = (s_ > 192) & (fr == 64)
    nblk[("fwd", int(np.where(big, np.ceil(m_ / 16), np.ceil(m_ / fr)).sum()))] = L
    rows_b = np.maximum(s_, 0 if leaf else 1)
    nblk[("bwd", int(np.ceil(rows_b / br).sum()))] = L

rows = list(csv.DictReader(open(sys.argv[1])))
agg = {}
for r in rows:
    m = re.search(r"(k_(?:fwd|bwd)(?:_mix|_rows)?)<(\d+)", r["Kernel_Name"])
    if not m:
        continue
    name, P = m.group(1), int(m.group(2))
    g = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
    agg.setdefault((name, P, g), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

out = []
for (name, P, g), d in sorted(agg.items(), key=lambda kv: (kv[0][0][:5], kv[0][1], -nblk.get((kv[0][0][2:5], kv[0][2]), -1))):
    L = nblk.get((name[2:5], g), -1)
    us = np.mean(d) / 1e3
    byt = mat[L] + P * vec[L] if L >= 0 else float("nan")
    out.append({"kernel": name, "P": P, "workgroups": g, "level": L, "fronts": int(cnt[L]) if L >= 0 else None, "launches": len(d),
                "avg_us": round(us, 2), "MB": round(byt / 1e6, 2), "GBps": round(byt / us / 1e3, 1)})
    print(f"{name:10s} P={P} workgroups={g:6d} level={L:2d} fronts={cnt[L] if L >= 0 else 0:5d} n={len(d):5d} "
          f"avg={us:7.2f} us  {byt / 1e6:7.2f} MB  {byt / us / 1e3:8.1f} GB/s")
for name in ("k_fwd", "k_bwd"):
    for P in (1, 4):
        sel = [o for o in out if o["kernel"].startswith(name) and o["P"] == P and o["level"] >= 0]
        if sel:
            us = sum(o["avg_us"] for o in sel)
            mb = sum(o["MB"] for o in sel)
            print(f"{name}* P={P}: one sweep = {us:.1f} us over {len(sel)} levels, {mb:.1f} MB -> {mb / us * 1e3:.0f} GB/s")
json.dump(out, open("gpurun_out/level_roofline.json", "w"), indent=1)
