"""Which items of the 64-solve sweep trip the a-posteriori guard (perturbed pivots / residual above the bound)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging
logging.getLogger("pl_v18.solver_fem").setLevel(logging.ERROR)
from pl_fem_vectoriel_amd.sweep import multiband_sweep_items
from pl_fem_vectoriel_amd.mesh import generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver
items = multiband_sweep_items()
meshes = {}
bad = 0
worst = 0.0
for it in items:
    g = it.geometry()
    if it.mesh_key not in meshes:
        meshes = {it.mesh_key: generate_mesh(g, it.mesh_refinement, it.mesh_levels)}
        solver = TrueVectorialMaxwellSolver(g, device=0, eig_tol=float(os.environ.get('EIG_TOL', '1e-10')))
    solver.geometry, solver.k0 = g, g.k0
    t0 = time.perf_counter()
    modes = solver.solve_vectorial_modes(meshes[it.mesh_key], it.n_modes)
    st = solver.last_stats
    flag = st["refined"] or st["pivot_perturbations"] > 0
    bad += flag
    if not flag:
        worst = max(worst, st['true_residual_first'])
    if flag or "-v" in sys.argv:
        print(f"{it.index:2d} {it.arrangement:24s} pitch {it.pitch_um:4.1f} lam {it.wavelength_um:.2f} N {st['N']:6d} perturbed {st['pivot_perturbations']} "
              f"res first {st['true_residual_first']:.2e} final {st['true_residual']:.2e} refined {st['refined']} {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
print("items that needed the guard:", bad, "of", len(items), " largest first-pass residual of the others:", worst)
