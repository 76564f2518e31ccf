"""Cold C1 solve time against the leaf size of the front tree (leaf_elems: 8 / 16 / 32 give 13 / 12 / 11 levels at C1)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging
logging.disable(logging.WARNING)
import numpy as np, torch
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver

levels = int(sys.argv[1]) if len(sys.argv) > 1 else 1
geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(geom, 1.0, levels)
for le in (8, 16, 32, 64):
    ts, st = [], None
    for rep in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solver = TrueVectorialMaxwellSolver(geom, device=0, leaf_elems=le, reuse_symbolic=False)
        modes = solver.solve_vectorial_modes(mesh, 10)
        ts.append(time.perf_counter() - t0)
        st = solver.last_stats
    print(f"leaf_elems {le:3d}: cold solve {1e3 * np.median(ts[2:]):7.2f} ms  symbolic {1e3 * st['t_symbolic']:.2f} context {1e3 * st['t_context']:.2f} "
          f"factor {st['factor_us'] / 1e3:.2f} lanczos {st['lanczos_us'] / 1e3:.2f}" +
          f"  n_opinv {st.get('n_opinv')} res {st['true_residual']:.1e} perturbed {st['pivot_perturbations']}", flush=True)
