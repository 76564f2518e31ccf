"""Lanczos basis-size experiment at C1: block solves, restarts and wall time of plfem_lanczos_shift_invert vs ncv
(NCVS=104,132,160 TOL=1e-8 to choose the list and the tolerance)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh, _native
from pl_fem_vectoriel_amd.solver_fem import _core_table, shift_estimate
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 1
k = int(sys.argv[2]) if len(sys.argv) > 2 else 22
g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(g, 1.0, levels)
sym = _native.Symbolic(mesh.p, mesh.t)
ctx = _native.Context(sym, 0, max_ncv=160)
ctx.assemble(_core_table(g), g.n_core ** 2, g.n_clad ** 2, g.k0, 1.0)
sigma = shift_estimate(g)
ctx.factor(sigma)
ref = None
ncvs = [int(v) for v in os.environ["NCVS"].split(",")] if os.environ.get("NCVS") else (2 * k + 1, 48, 52, 56, 60, 64, 72, 80, 96, 128)
tol = float(os.environ.get("TOL", "1e-10"))
for ncv in ncvs:
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ev, V, st = ctx.lanczos(k, ncv, tol, 12000, sigma)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    if ref is None: ref = ev.copy()
    print(f"ncv={ncv:4d}  block_solves={st.get('n_block_solves')}  n_opinv={st['n_opinv']}  restarts={st['restarts']}  "
          f"wall={best * 1e3:7.2f} ms  max|dlam|/lam={np.abs(ev - ref).max() / np.abs(ref).max():.1e}", flush=True)
