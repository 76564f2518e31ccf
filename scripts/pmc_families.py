#!/usr/bin/env python3
"""HBM traffic and MFMA counters per kernel FAMILY of one bench run, from separate rocprofv3 --pmc passes.

usage: pmc_families.py <fetch_counter_collection.csv> <write_counter_collection.csv> <mfma_counter_collection.csv|-> <out.json>

FETCH_SIZE / WRITE_SIZE are KiB per dispatch.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half
the bytes of a wide coalesced streaming read, so the read side is doubled (`hbm_bytes = 2 FETCH + WRITE`); the kernels
here stream 8-byte lanes (512-B wave requests), a width the guide calls uncalibrated: raw values are kept beside the
corrected ones.  Every family is normalised by the number of times its unit occurs in the run (a block solve, an SpMV,
an assembly, a factorisation), counted from the dispatches of one marker kernel of that unit.
MFMA pass (factor kernels): SQ_INSTS_VALU_MFMA_MOPS_F64 (512 flop each), SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
summed over the dispatches of a family."""
import collections
import csv
import json
import re
import sys

FAMILIES = [
    # name, kernel regex, marker regex (one dispatch per unit), unit
    ("forward sweep (k_fwd, k_fwd_mix, k_fwd_rows)", r"k_fwd(_mix|_rows)?<4", r"k_permute_out<4|k_permute_dot_first<4", "block solve (P = 4)"),
    ("backward sweep (k_bwd, k_bwd_rows)", r"k_bwd(_rows)?<4", r"k_permute_out<4|k_permute_dot_first<4", "block solve (P = 4)"),
    ("k_fwd / k_fwd_mix (tile-form forward levels)", r"k_fwd(_mix)?<4", None, "launch"),
    ("k_bwd_rows", r"k_bwd_rows<4", None, "launch"),
    ("k_bwd (leaf level)", r"k_bwd<4", None, "launch"),
    ("k_spmv_b_block", r"k_spmv_b_block", None, "launch"),
    ("assembly (k_element_matrices + k_csr_gather)", r"k_element_matrices|k_csr_gather", r"k_csr_gather", "assembly"),
    ("k_ldl_first_panel", r"k_ldl_first_panel", None, "launch"),
    ("k_ldl_update", r"k_ldl_update", None, "launch"),
    ("factorisation (all kernels)", r"k_ldl_|k_front_gather|k_leaf_assemble|k_form_z|k_mirror_z|k_mirror_x", r"k_leaf_assemble", "factorisation"),
]


def load(path, counters):
    per = collections.defaultdict(lambda: collections.defaultdict(float))      # dispatch -> counter -> value
    name = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") not in counters:
            continue
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
    return per, name


def totals(per, name, pattern, counter):
    pat = re.compile(pattern)
    ids = [d for d, n in name.items() if pat.search(n)]
    return sum(per[d].get(counter, 0.0) for d in ids), len(ids)


def matched_names(name, pattern):
    """Distinct kernel names a family's regex matched (argument lists cut off): bench.py refuses the file when one of them
    is no longer in the built library (the counters were then collected on other code)."""
    pat = re.compile(pattern)
    return sorted({re.sub(r"^void ", "", n).replace("plfem::(anonymous namespace)::", "").split("(")[0] for n in name.values() if pat.search(n)})


def main():
    fetch_csv, write_csv, mfma_csv, out = sys.argv[1:5]
    fper, fname = load(fetch_csv, {"FETCH_SIZE"})
    wper, wname = load(write_csv, {"WRITE_SIZE"})
    res = {"note": "hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md, HBM section); per unit",
           "families": {}}
    for fam, pat, marker, unit in FAMILIES:
        f, nf = totals(fper, fname, pat, "FETCH_SIZE")
        w, nw = totals(wper, wname, pat, "WRITE_SIZE")
        if nf == 0 or nw == 0:
            continue
        uf = totals(fper, fname, marker, "FETCH_SIZE")[1] if marker else nf
        uw = totals(wper, wname, marker, "WRITE_SIZE")[1] if marker else nw
        res["families"][fam] = {"regex": pat, "unit": unit, "dispatches_fetch_pass": nf, "dispatches_write_pass": nw,
                                "units_fetch_pass": uf, "units_write_pass": uw,
                                "fetch_bytes_raw": f * 1024 / uf, "write_bytes": w * 1024 / uw,
                                "hbm_bytes": 2 * f * 1024 / uf + w * 1024 / uw,
                                "kernel_names": matched_names(fname, pat)}
    if mfma_csv != "-":
        names = {"SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_MFMA"}
        mper, mname = load(mfma_csv, names)
        res["mfma"] = {}
        for fam, pat, marker, unit in FAMILIES:
            if not re.search("ldl|factor", fam):
                continue
            row = {}
            for c in sorted(names):
                v, n = totals(mper, mname, pat, c)
                if n:
                    row[c] = v
                    row["dispatches"] = n
            if row:
                units = totals(mper, mname, marker, "SQ_BUSY_CYCLES")[1] if marker else row["dispatches"]
                row["units"] = units
                if "SQ_INSTS_VALU_MFMA_MOPS_F64" in row:
                    row["mfma_flop_per_unit"] = 512.0 * row["SQ_INSTS_VALU_MFMA_MOPS_F64"] / max(units, 1)
                if "SQ_VALU_MFMA_BUSY_CYCLES" in row and row.get("SQ_BUSY_CYCLES"):
                    row["mfma_busy_over_sq_busy"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / row["SQ_BUSY_CYCLES"]
                res["mfma"][fam] = row
    json.dump(res, open(out, "w"), indent=1)
    for fam, v in res["families"].items():
        print(f"{fam:50s} {v['hbm_bytes'] / 1e6:10.2f} MB per {v['unit']}")
    for fam, v in res.get("mfma", {}).items():
        print(f"{fam:50s} " + "  ".join(f"{k}={val:.4g}" for k, val in v.items()))


if __name__ == "__main__":
    main()
