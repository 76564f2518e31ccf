#!/usr/bin/env python3
"""Front-by-front comparison of the HIP multifrontal LDL^T with the NumPy emulation
(tests/front_emulation.py) driven by the same symbolic arrays.  Debugging aid; run on a GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh, _native
from pl_fem_vectoriel_amd.solver_fem import shift_estimate, _core_table
from oracle.p2 import MeshTriLite, P2Basis
from oracle import hfield
import front_emulation as fe

ref = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0)
mesh = generate_mesh(g, ref, 0)
sym = _native.Symbolic(mesh.p, mesh.t, leaf_elems=24)
ctx = _native.Context(sym, 0, max_ncv=65)
ctx.assemble(_core_table(g), g.n_core**2, g.n_clad**2, g.k0, 1.0)
om = MeshTriLite(mesh.p, mesh.t); basis = P2Basis(om)
em = hfield.element_matrices(g, basis)
sigma = shift_estimate(g)
Ke = fe.element_K(em, g.k0**2, sigma)
T = fe.FrontTree(sym)
ne, N = mesh.nelements, sym.N
eb = ctx.debug_copy("elem", 0, ne * 288).reshape(ne, 8, 6, 6)
Axx = em['kxx'] + em['div_xx'] - g.k0**2 * em['mass']
print("elem Axx max rel diff", np.abs(eb[:, 0] - Axx).max() / np.abs(Axx).max(), flush=True)

def gpu_front(f):
    m = T.m(f)
    return T.device_front(ctx, f, with_schur=True)   # [row, col]; the Schur complement while its level arena is alive

def cmp(name, a, b, tol=1e-9):
    sc = max(np.abs(b).max(), 1e-300)
    err = np.abs(a - b).max() / sc
    print(f"{name}: max rel err {err:.3e} (scale {sc:.3e}){'' if err < tol else '   <<<<<< MISMATCH'}", flush=True)
    return err

Fs, Ds = fe.factor(T, Ke)
# staged: leaf level assembled
ctx.debug_factor_until(sigma, T.L, 0, 0)
S = [None] * T.nf
for f in (T.leaf0, T.leaf0 + 1, T.nf - 1):
    cmp(f"leaf {f} assembled (m={T.m(f)}, s2={T.s2(f)})", gpu_front(f), fe.assemble_front(T, f, Ke, S))
ctx.factor(sigma); ctx.synchronize()
print(ctx.timings(), flush=True)
worst = 0.0
for f in range(T.nf - 1, -1, -1):
    if f in (0, 1, 2, 3, 4, 7, 8, T.leaf0 - 1, T.leaf0, T.nf - 1) or f % 61 == 0:
        s2, m = T.s2(f), T.m(f)
        Fg = gpu_front(f)
        e1 = cmp(f"front {f} (m={m}, s2={s2}) F11", Fg[:s2, :s2], Fs[f][:s2, :s2]) if s2 else 0
        e2 = cmp(f"front {f} L21", Fg[s2:, :s2], Fs[f][s2:, :s2]) if s2 and m > s2 else 0
        e3 = cmp(f"front {f} L21^T", Fg[:s2, s2:], Fs[f][:s2, s2:]) if s2 and m > s2 else 0
        e4 = cmp(f"front {f} S", Fg[s2:, s2:], Fs[f][s2:, s2:]) if m > s2 else 0
        dg = ctx.debug_copy("delta", 2 * T.fptr[f], s2)
        e5 = cmp(f"front {f} D", dg, Ds[f]) if s2 else 0
        worst = max(worst, e1, e2, e3, e4, e5)
print("worst", worst, flush=True)
# solve sweeps on the emulated factors vs HIP
rng = np.random.default_rng(0)
interior = sym.array('interior'); idx = np.concatenate([interior, interior + N])
rhs = np.zeros(2 * N); rhs[idx] = rng.standard_normal(len(idx))
x = fe.solve(T, Fs, Ds, rhs)
xg = ctx.solve(torch.from_numpy(rhs).cuda(), 0).cpu().numpy()
print("solve: |x_gpu - x_emul| / |x_emul| =", np.linalg.norm(xg - x) / np.linalg.norm(x), flush=True)
A, B, _, _, _, _, _ = hfield.assemble_hfield_system_fused(g, om)
A_int, B_int, _ = hfield.restrict_interior(A, B, basis)
K = (A_int - sigma * B_int).tocsc()
for rs in (0, 1, 2):
    xg = ctx.solve(torch.from_numpy(rhs).cuda(), rs).cpu().numpy()
    print(f"refine={rs}: residual {np.linalg.norm(K @ xg[idx] - rhs[idx]) / np.linalg.norm(rhs[idx]):.3e}", flush=True)
