"""Smallest pivots of the LDL^T of one sweep item: which front, which block step, how they compare with the rest of the block."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging
logging.getLogger("pl_v18.solver_fem").setLevel(logging.ERROR)
import numpy as np
from pl_fem_vectoriel_amd.sweep import multiband_sweep_items
from pl_fem_vectoriel_amd.mesh import generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver

idx = int(sys.argv[1]) if len(sys.argv) > 1 else 13
it = multiband_sweep_items()[idx]
g = it.geometry()
mesh = generate_mesh(g, it.mesh_refinement, it.mesh_levels)
solver = TrueVectorialMaxwellSolver(g, device=0)
solver.reuse_symbolic = True
modes = solver.solve_vectorial_modes(mesh, it.n_modes)
st = solver.last_stats
print(it.arrangement, it.pitch_um, it.wavelength_um, "perturbed", st["pivot_perturbations"], "res", st["true_residual_first"], "sigma", st.get("sigma"))
ent = list(solver._cache.values())[0]
sym, ctx = ent["sym"], ent["ctx"]
fs, fb, fptr = sym.array("fs"), sym.array("fb"), sym.array("fnode_ptr")
dpn = 2
delta = ctx.debug_copy("delta", 0, dpn * int(fptr[-1]))
rows = []
for f in range(len(fs)):
    d = delta[dpn * fptr[f]: dpn * fptr[f] + dpn * fs[f]]
    if len(d) == 0:
        continue
    a = np.abs(d)
    k = int(np.argmin(a))
    rows.append((a[k], f, k, len(d), dpn * int(fb[f]), float(np.median(a)), float(a.max())))
rows.sort()
for r in rows[:12]:
    lvl = int(np.floor(np.log2(r[1] + 1)))
    print(f"|d| {r[0]:.3e} front {r[1]} (depth {lvl}) pivot {r[2]} of s1 {r[3]} s2 {r[4]} median|d| {r[5]:.3e} max|d| {r[6]:.3e}")
f = rows[0][1]
d = delta[dpn * fptr[f]: dpn * fptr[f] + dpn * fs[f]]
k = rows[0][2]
b0 = (k // 32) * 32
np.set_printoptions(linewidth=200, precision=3)
print("block", b0, d[b0:b0 + 32])
print("negative pivots total", int((delta[delta != 1.0] < 0).sum()), "of", int((delta != 1.0).sum()))
