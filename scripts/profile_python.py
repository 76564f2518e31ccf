"""Host-side overhead hunting at C1: wall time of a cold solve vs its accounted phases, and the cost of
tearing the previous solver / mode list down (what bench.py's loop pays between steps)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver
g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(g, 1.0, 1)
def once():
    s = TrueVectorialMaxwellSolver(g, device=0, reuse_symbolic=False)
    m = s.solve_vectorial_modes(mesh, 10)
    return s, m
for _ in range(3): s, m = once()
for rep in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s2, m2 = once()
    t1 = time.perf_counter()
    st = s2.last_stats
    acc = st['t_symbolic'] + st['t_context'] + st['t_device'] + st['t_copy_out']
    ta = time.perf_counter(); del m; tb = time.perf_counter(); del s; tc = time.perf_counter()
    print(f"solve wall {1e3*(t1-t0):.2f} ms  t_total {1e3*st['t_total']:.2f}  accounted {1e3*acc:.2f} | del modes {1e3*(tb-ta):.2f} ms  del solver {1e3*(tc-tb):.2f} ms")
    s, m = s2, m2
pr = cProfile.Profile(); pr.enable(); s2, m2 = once(); del m; del s; pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
