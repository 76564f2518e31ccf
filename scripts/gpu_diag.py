#!/usr/bin/env python3
"""Diagnostic run on a GPU box: stage-by-stage comparison of the HIP path against the oracle."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import scipy.sparse as sp
import scipy.sparse.linalg as spla
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
from pl_fem_vectoriel_amd import _native
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver, shift_estimate, _core_table
from oracle.p2 import MeshTriLite
from oracle import hfield

L = int(sys.argv[1]) if len(sys.argv) > 1 else 0
do_oracle_eig = (len(sys.argv) <= 2) or sys.argv[2] != "noeig"
g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
mesh = generate_mesh(g, 1.0, L)
print("mesh", mesh.nvertices, mesh.nelements, flush=True)
t0 = time.time(); sym = _native.Symbolic(mesh.p, mesh.t); t1 = time.time()
print("symbolic %.1f ms" % ((t1 - t0) * 1e3), sym.info, flush=True)
t0 = time.time(); ctx = _native.Context(sym, 0, max_ncv=65); torch.cuda.synchronize(); t1 = time.time()
print("context %.1f ms" % ((t1 - t0) * 1e3), flush=True)
cores = _core_table(g)
ctx.assemble(cores, g.n_core**2, g.n_clad**2, g.k0, 1.0); ctx.synchronize()
t0 = time.time(); ctx.assemble(cores, g.n_core**2, g.n_clad**2, g.k0, 1.0); ctx.synchronize(); t1 = time.time()
print("assemble wall %.3f ms" % ((t1 - t0) * 1e3), flush=True)

# oracle matrices
om = MeshTriLite(mesh.p, mesh.t)
A, B, basis, Dxx, Dyy, Dxy, Minv = hfield.assemble_hfield_system_fused(g, om, eliminate_zeros=False)
N = basis.N
rowptr = sym.array("rowptr"); colind = sym.array("colind")
def blk(name): return sp.csr_matrix((ctx.block_values(name), colind, rowptr), shape=(N, N))
refs = {"Axx": A[:N, :N], "Axy": A[:N, N:], "Ayx": A[N:, :N], "Ayy": A[N:, N:], "Minv": Minv, "Dxx": Dxx, "Dxy": Dxy, "Dyy": Dyy}
for name, R in refs.items():
    G = blk(name)
    d = abs(G - R)
    scale = abs(R).max()
    # row-relative
    rowmax = np.maximum(abs(R).max(axis=1).toarray().ravel(), 1e-300)
    rel = (d.max(axis=1).toarray().ravel() / rowmax).max()
    print(f"block {name}: max abs diff {d.max():.3e} (scale {scale:.3e}) max row-relative {rel:.3e}", flush=True)

A_int, B_int, interior = hfield.restrict_interior(A, B, basis)
ns = len(interior)
idx = np.concatenate([interior, interior + N])
rng = np.random.default_rng(0)
xi = rng.standard_normal(2 * ns)
xfull = np.zeros(2 * N); xfull[idx] = xi
xd = torch.from_numpy(xfull).cuda()
for which, M in (("A", A_int), ("B", B_int)):
    y = ctx.spmv(which, xd).cpu().numpy()
    yref = M @ xi
    print(f"spmv {which}: rel err {np.abs(y[idx]-yref).max()/np.abs(yref).max():.3e}  boundary max {np.abs(np.delete(y, idx)).max():.3e}", flush=True)

sigma = shift_estimate(g)
ctx.factor(sigma); ctx.synchronize()
t0 = time.time(); ctx.factor(sigma); ctx.synchronize(); t1 = time.time()
print("factor wall %.3f ms" % ((t1 - t0) * 1e3), ctx.timings(), flush=True)
K = (A_int - sigma * B_int).tocsc()
b = rng.standard_normal(2 * ns)
bfull = np.zeros(2 * N); bfull[idx] = b
bd = torch.from_numpy(bfull).cuda()
for rs in (0, 1):
    x = ctx.solve(bd, rs); ctx.synchronize()
    t0 = time.time(); x = ctx.solve(bd, rs); ctx.synchronize(); t1 = time.time()
    xh = x.cpu().numpy()
    r = K @ xh[idx] - b
    print(f"solve refine={rs}: wall {1e3*(t1-t0):.3f} ms  rel residual {np.linalg.norm(r)/np.linalg.norm(b):.3e}  boundary max {np.abs(np.delete(xh, idx)).max():.3e}", flush=True)
if 2 * ns < 400000:
    t0 = time.time(); lu = spla.splu(K); t1 = time.time()
    xs = lu.solve(b)
    print(f"splu {t1-t0:.2f}s; gpu-vs-splu rel diff {np.linalg.norm(xh[idx]-xs)/np.linalg.norm(xs):.3e}; splu residual {np.linalg.norm(K@xs-b)/np.linalg.norm(b):.3e}", flush=True)

k = 22; ncv = 45
for tol in (1e-10,):
    t0 = time.time(); evals, evecs, st = ctx.lanczos(k, ncv, tol, 12000, sigma); ctx.synchronize(); t1 = time.time()
    print(f"lanczos tol={tol}: wall {1e3*(t1-t0):.2f} ms", st, ctx.timings(), flush=True)
print("evals", evals, flush=True)
ev = evecs.cpu().numpy()
# residuals of eigenpairs
V = ev[:, idx].T
res = [np.linalg.norm(A_int @ V[:, i] - evals[i] * (B_int @ V[:, i])) / np.linalg.norm(A_int @ V[:, i]) for i in range(k)]
print("max eigen residual", max(res), flush=True)
G = V.T @ (B_int @ V)
print("B-orthonormality", np.abs(G - np.eye(k)).max(), flush=True)
if do_oracle_eig:
    t0 = time.time()
    w, U = spla.eigsh(A_int, k=k, M=B_int, sigma=sigma, which="LM", tol=1e-7, maxiter=12000)
    t1 = time.time()
    o = np.argsort(w); w = w[o]; U = U[:, o]
    print(f"oracle eigsh {t1-t0:.2f}s", flush=True)
    print("max |dlambda|", np.abs(w - evals).max(), " max |dn_eff|", np.abs(np.sqrt(w) - np.sqrt(evals)).max() / g.k0, flush=True)
    errs = []
    for i in range(k):
        a = V[:, i] / np.linalg.norm(V[:, i]); bb = U[:, i] / np.linalg.norm(U[:, i])
        errs.append(min(np.linalg.norm(a - bb), np.linalg.norm(a + bb)))
    print("field L2 errs", np.array2string(np.array(errs), precision=2), flush=True)
# full solver
solver = TrueVectorialMaxwellSolver(g)
for rep in range(3):
    solver.clear_cache()
    t0 = time.time(); modes = solver.solve_vectorial_modes(mesh, 10); t1 = time.time()
    print(f"full solve (cold) {1e3*(t1-t0):.1f} ms  modes {len(modes)}", {k_: (round(v, 4) if isinstance(v, float) else v) for k_, v in solver.last_stats.items()}, flush=True)
for rep in range(3):
    t0 = time.time(); modes = solver.solve_vectorial_modes(mesh, 10); t1 = time.time()
    print(f"full solve (warm) {1e3*(t1-t0):.1f} ms  modes {len(modes)}", flush=True)
print([round(m["n_eff"], 6) for m in modes])
print({k_: v for k_, v in modes[0].items() if not hasattr(v, "shape")})
