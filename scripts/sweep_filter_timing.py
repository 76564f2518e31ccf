"""Timing experiment: the sweeps of a C1 block solve with all fronts / without the fronts of more than 128 owned DOFs /
with only those (plfem_debug_solve_block).  Run under rocprofv3 --kernel-trace; the phases are separated by marker
launches of k_permute_in (one per solve)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh, _native
from pl_fem_vectoriel_amd.solver_fem import _core_table, shift_estimate
g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(g, 1.0, 1)
sym = _native.Symbolic(mesh.p, mesh.t)
ctx = _native.Context(sym, 0, max_ncv=65)
ctx.assemble(_core_table(g), g.n_core ** 2, g.n_clad ** 2, g.k0, 1.0)
ctx.factor(shift_estimate(g))
for flt in (0, 1, 2):
    ctx.debug_solve_block(5, flt)
torch.cuda.synchronize()
