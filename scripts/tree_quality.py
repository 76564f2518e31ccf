"""Front-tree quality of the symbolic analysis on the benchmark meshes (CPU only): factor flops, solve entries,
front storage, largest front, block-step critical path vs level-synchronous step count, host time."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh, _native

def report(name, mesh, leaf=0):
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); sym = _native.Symbolic(mesh.p, mesh.t, leaf_elems=leaf); best = min(best, time.perf_counter() - t)
    fs, fb = sym.array("fs"), sym.array("fb")
    nf, L = len(fs), sym.info["levels"]
    steps = (2 * fs + 31) // 32
    cp = np.zeros(nf, int)
    for f in range(nf - 1, -1, -1):
        c = max(cp[2 * f + 1], cp[2 * f + 2]) if 2 * f + 2 < nf else 0
        cp[f] = steps[f] + c
    lvl = sum(int(steps[(1 << l) - 1:(1 << (l + 1)) - 1].max()) for l in range(L + 1))
    i = sym.info
    print(f"{name:14s} N={i['N']:7d} L={L:2d} flops={i['factor_flops']/1e9:8.2f}G solve={i['solve_entries']/1e6:7.1f}M "
          f"front={i['front_doubles']/1e6:7.1f}M max_m={i['max_front']:5d} cp={cp[0]:4d} lvl_steps={lvl:4d} host={best*1e3:6.1f}ms")
    return sym

if __name__ == "__main__":
    g7 = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    g19 = MCFGeometry(19, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    which = sys.argv[1:] or ["L0", "C1", "L2", "C5"]
    if "L0" in which: report("7-core L0", generate_mesh(g7, 1.0, 0))
    if "C1" in which: report("7-core L1 (C1)", generate_mesh(g7, 1.0, 1))
    if "L2" in which: report("7-core L2", generate_mesh(g7, 1.0, 2))
    if "C5" in which: report("19-core L2 (C5)", generate_mesh(g19, 1.0, 2))


def kary_top(sym, depth):
    """What 2^depth-way splits at the top of the tree would do, from the binary tree at hand: the
    separators of levels 1 .. depth - 1 join the root's in ONE front (children of a k-way node share one separator front);
    everything below keeps its fronts.  Returns launches per sweep, level-synchronous block steps, factor flops, solve
    entries of the modified tree (same conventions as report())."""
    fs, fb = 2 * sym.array("fs").astype(np.int64), 2 * sym.array("fb").astype(np.int64)     # DOFs
    nf, L = len(fs), sym.info["levels"]
    steps = (fs + 31) // 32
    merged = int(sum(fs[(1 << l) - 1:(1 << (l + 1)) - 1].sum() for l in range(depth)))
    lvl_steps = (merged + 31) // 32 + sum(int(steps[(1 << l) - 1:(1 << (l + 1)) - 1].max()) for l in range(depth, L + 1))
    flops = float(merged) ** 3 + float(sum(fs[f] * (fs[f] + fb[f]) ** 2 for f in range((1 << depth) - 1, nf)))
    entries = merged * merged + int(sum(fs[f] * (fs[f] + 2 * fb[f]) for f in range((1 << depth) - 1, nf)))
    return {"levels": L + 2 - depth, "launches_per_sweep_pair": 2 * (L + 2 - depth), "lvl_steps": int(lvl_steps), "top_front": merged,
            "flops_G": flops / 1e9, "solve_entries_M": entries / 1e6}


if __name__ == "__main__" and "KARY" in sys.argv[1:]:
    g7 = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    sym = _native.Symbolic(*(lambda m: (m.p, m.t))(generate_mesh(g7, 1.0, 1)))
    for d in (1, 2, 3, 4):
        print(f"top {2 ** d:2d}-way (levels 0..{d - 1} in one front):", kary_top(sym, d))
