"""cProfile of one cold solve_vectorial_modes at C1 (host-side overhead hunting)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver
g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(g, 1.0, 1)
def once():
    s = TrueVectorialMaxwellSolver(g, device=0)
    m = s.solve_vectorial_modes(mesh, 10)
    return s
for _ in range(3): once()
pr = cProfile.Profile(); pr.enable(); s = once(); pr.disable()
st = s.last_stats
print({k: round(v * 1e3, 3) for k, v in st.items() if k.startswith('t_')})
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
