"""Where does the analysis lose ~1 ms inside the solve loop?  Same build timed (A) alone, (B) after HIP is initialised,
(C) with a device context alive, (D) alternating with full solves."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging
logging.disable(logging.WARNING)
import numpy as np
from pl_fem_vectoriel_amd import _native, MCFGeometry, generate_mesh
geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(geom, 1.0, 1)

def builds(tag, n=25, between=None):
    ts = []
    for rep in range(n):
        if between:
            between()
        t0 = time.perf_counter()
        s = _native.Symbolic(mesh.p, mesh.t)
        ts.append(time.perf_counter() - t0)
    print(f"{tag:46s} median {1e3 * np.median(ts[3:]):.2f} ms  min {1e3 * min(ts):.2f}", flush=True)
    return s

builds("A alone")
import torch
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
builds("B HIP initialised")
s = builds("B again")
ctx = _native.Context(s, 0, max_ncv=132)
builds("C one device context alive")
ctx.close()
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver
solver = TrueVectorialMaxwellSolver(geom, device=0)
solver.solve_vectorial_modes(mesh, 10)
builds("D warm solve (same solver) between builds", between=lambda: solver.solve_vectorial_modes(mesh, 10))
def cold():
    sv = TrueVectorialMaxwellSolver(geom, device=0, reuse_symbolic=False)
    sv.solve_vectorial_modes(mesh, 10)
    cold.t.append(sv.last_stats["t_symbolic"])
cold.t = []
builds("E cold solve between builds", between=cold)
print("   t_symbolic inside those cold solves: median %.2f ms" % (1e3 * np.median(cold.t[3:])))
builds("F alone again")
