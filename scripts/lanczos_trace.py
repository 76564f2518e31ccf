"""Convergence history of the block Lanczos run at C1 (PLFEM_LANCZOS_TRACE): columns, converged pairs, largest relative
residual of the wanted pairs after every block step."""
import sys, os
os.environ["PLFEM_LANCZOS_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging; logging.disable(logging.WARNING)
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver
levels = int(sys.argv[1]) if len(sys.argv) > 1 else 1
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-10
geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(geom, 1.0, levels)
s = TrueVectorialMaxwellSolver(geom, device=0, eig_tol=tol)
s.solve_vectorial_modes(mesh, 10)
print({k: s.last_stats[k] for k in ("n_opinv", "nconv", "true_residual", "lanczos_us")})
