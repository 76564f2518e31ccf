"""Host-side overhead hunting at C1: where the wall time of bench.py's cold-step loop goes beyond the phases
solve_vectorial_modes accounts for (construction, teardown of the previous solver / mode list)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver
g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(g, 1.0, 1)
solver = modes = None
for it in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s = TrueVectorialMaxwellSolver(g, device=0, reuse_symbolic=False)
    t1 = time.perf_counter()
    m = s.solve_vectorial_modes(mesh, 10)
    t2 = time.perf_counter()
    old_modes, modes = modes, m
    del old_modes
    t3 = time.perf_counter()
    ent = list(solver._cache.values()) if solver is not None else []
    old_solver, solver = solver, s
    if old_solver is not None:
        c = old_solver.__dict__.get("_last_ent")
    del old_solver
    t4 = time.perf_counter()
    st = s.last_stats
    acc = st['t_symbolic'] + st['t_context'] + st['t_pinned'] + st['t_call']
    print({k: round(1e3 * st[k], 2) for k in ('t_symbolic', 't_context', 't_workspace', 't_pinned', 't_call')},
          {k: round(st[k] / 1e3, 2) for k in ('upload_us', 'assemble_us', 'factor_us', 'lanczos_us', 'post_us', 'residual_us', 'call_us')}, end="  ")
    print(f"construct {1e3*(t1-t0):.2f}  solve {1e3*(t2-t1):.2f} (accounted {1e3*acc:.2f})  free modes {1e3*(t3-t2):.2f}  free solver {1e3*(t4-t3):.2f}  total {1e3*(t4-t0):.2f} ms")
