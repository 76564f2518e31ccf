"""CPU only: NumPy emulation of the multifrontal LDL^T on one sweep item with (a) scalar pivots in the static order
(round 2: vanishing pivots perturbed to 1e-13 of their 32 x 32 block) and (b) 2 x 2 node-pair pivots (Hx, Hy of one
P2 node = adjacent local DOFs 2q, 2q+1; D block diagonal, no permutation).  Prints the smallest pivots either way and
the accuracy of K^-1 b against SuperLU.   usage: pair_pivot_emulation.py <item index> [mesh levels]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scipy.sparse.linalg as spla

import front_emulation as fe
from oracle import hfield
from oracle.p2 import MeshTriLite, P2Basis
from pl_fem_vectoriel_amd import _native
from pl_fem_vectoriel_amd.mesh import generate_mesh
from pl_fem_vectoriel_amd.solver_fem import shift_estimate
from pl_fem_vectoriel_amd.sweep import multiband_sweep_items

NB = 32
ROW_THR = os.environ.get('ROW_THR', '1') == '1'   # threshold relative to the pair's own rows instead of the block


def ldl_scalar(Fm, s2, stats):
    F = Fm.copy()
    d = np.zeros(s2)
    thr = 0.0
    for k in range(s2):
        if k % NB == 0:
            blk = F[k:min(k + NB, s2), k:min(k + NB, s2)]
            thr = max(1e-13 * np.abs(blk).max(), 1e-300)
        dk = F[k, k]
        if not abs(dk) >= thr:
            dk = -thr if dk < 0 else thr
            stats["perturbed"] += 1
        stats["rel"].append(abs(dk) / (thr * 1e13))
        d[k] = dk
        l = F[k + 1:, k] / dk
        stats["growth"] = max(stats["growth"], float(np.abs(l).max()) if len(l) else 0.0)
        F[k + 1:, k + 1:] -= np.outer(l, F[k + 1:, k])
        F[k + 1:, k] = l
    return F, d


def ldl_pairs(Fm, s2, stats):
    """Node-pair pivots as kernels_front.hip takes them: two scalar pivots in the static order or one 2 x 2 pivot through the
    explicit inverse, whichever amplifies rounding errors less; vanishing = below 1e-13 of the pair's own rows."""
    F = Fm.copy()
    Dinv = np.zeros((s2, 2))       # [:, 0] diagonal of D^-1, [:, 1] the off-diagonal entry of the pair
    for k in range(0, s2, 2):
        if k % NB == 0:
            blk = F[k:min(k + NB, s2), k:min(k + NB, s2)]
            rowmax = np.abs(blk).max(axis=1)
            bmax = np.abs(blk).max()
        a, b, c = F[k, k], F[k + 1, k], F[k + 1, k + 1]
        thr = max(1e-13 * (max(rowmax[k % NB], rowmax[k % NB + 1]) if ROW_THR else bmax), 1e-300)
        det = a * c - b * b
        s = max(abs(a), abs(b), abs(c))
        if b * b * abs(det) <= a * a * s * s:
            stats["modes"][1] += 1
            for j in (k, k + 1):
                d = F[j, j]
                stats["rel"].append(abs(d) / (thr * 1e13))
                if not abs(d) >= thr:
                    d = -thr if d < 0 else thr
                    stats["perturbed"] += 1
                l = F[j + 1:, j] / d
                gr = float(np.abs(l[(k + 1 - j):]).max()) if len(l) > k + 1 - j else 0.0      # (not the in-pair multiplier)
                if gr > stats["growth"]:
                    stats["growth"] = gr
                    stats["where"] = (k, s2, F.shape[0], a, b, c)
                F[j + 1:, j + 1:] -= np.outer(l, F[j + 1:, j])
                F[j + 1:, j] = l
                Dinv[j] = (1.0 / d, 0.0)
        else:
            stats["modes"][0] += 1
            stats["rel"].append(abs(det) / s / (thr * 1e13))
            e11, e12, e22 = c / det, -b / det, a / det
            Dinv[k] = (e11, e12)
            Dinv[k + 1] = (e22, e12)
            C = F[k + 2:, k:k + 2].copy()
            Lc = np.stack([C[:, 0] * e11 + C[:, 1] * e12, C[:, 0] * e12 + C[:, 1] * e22], 1)
            gr = float(np.abs(Lc).max()) if len(Lc) else 0.0
            if gr > stats["growth"]:
                stats["growth"] = gr
                stats["where"] = (k, s2, F.shape[0], a, b, c)
            F[k + 2:, k + 2:] -= Lc @ C.T
            F[k + 2:, k:k + 2] = Lc
            F[k + 1, k] = 0.0
    return F, Dinv


def finish(F, s2):
    import scipy.linalg as sla
    L11 = np.tril(F[:s2, :s2], -1) + np.eye(s2)
    X = sla.solve_triangular(L11, np.eye(s2), lower=True, unit_diagonal=True) if s2 else np.zeros((0, 0))
    out = F.copy()
    out[:s2, :s2] = np.tril(X) + np.tril(X, -1).T
    Z = F[s2:, :s2] @ np.tril(X)
    out[s2:, :s2] = Z
    out[:s2, s2:] = Z.T
    return out


def factor(T, Ke, mode, stats):
    Fs, Ds, S = [None] * T.nf, [None] * T.nf, [None] * T.nf
    for f in range(T.nf - 1, -1, -1):
        Fm = fe.assemble_front(T, f, Ke, S)
        s2 = T.s2(f)
        F, D = (ldl_scalar if mode == "scalar" else ldl_pairs)(Fm, s2, stats)
        Fs[f], Ds[f] = finish(F, s2), D
        S[f] = Fs[f][s2:, s2:]
    return Fs, Ds


def solve(T, Fs, Ds, rhs, mode):
    N = T.N
    W, Y = [None] * T.nf, [None] * T.nf
    for f in range(T.nf - 1, -1, -1):
        mn = int(T.fs[f] + T.fb[f])
        m, s2 = 2 * mn, T.s2(f)
        fn = T.nodes(f)
        w = np.zeros(m)
        node = np.repeat(fn, 2)
        comp = np.tile([0, 1], mn)
        own = (np.arange(m) < s2) & (node >= 0)
        w[own] = rhs[comp[own] * N + node[own]]
        if f < T.leaf0:
            for ch, ci in ((2 * f + 1, T.c0), (2 * f + 2, T.c1)):
                inv = np.repeat(ci[T.fptr[f]:T.fptr[f] + mn], 2)
                ok = inv >= 0
                w[ok] += W[ch][T.s2(ch) + 2 * inv[ok] + comp[ok]]
        F = Fs[f]
        r = w[:s2].copy()
        t = np.tril(F[:s2, :s2]) @ r
        if mode == "scalar":
            ys = t / Ds[f]
        else:
            ys = Ds[f][:, 0] * t + Ds[f][:, 1] * t.reshape(-1, 2)[:, ::-1].ravel()
        w[s2:] -= F[s2:, :s2] @ r
        W[f], Y[f] = w, ys
    x = np.zeros(2 * N)
    for f in range(T.nf):
        mn = int(T.fs[f] + T.fb[f])
        m, s2 = 2 * mn, T.s2(f)
        fn = T.nodes(f)
        node = np.repeat(fn, 2)
        comp = np.tile([0, 1], mn)
        xb = np.where(node[s2:] >= 0, x[comp[s2:] * N + np.maximum(node[s2:], 0)], 0.0)
        F = Fs[f]
        v = np.concatenate([Y[f], -xb])
        xo = np.tril(F[:, :s2]).T @ v
        ok = node[:s2] >= 0
        x[comp[:s2][ok] * N + node[:s2][ok]] = xo[ok]
    return x


def main():
    idx = int(sys.argv[1]) if len(sys.argv) > 1 else 13
    levels = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    it = multiband_sweep_items()[idx]
    g = it.geometry()
    mesh = generate_mesh(g, it.mesh_refinement, levels)
    sym = _native.Symbolic(mesh.p, mesh.t)
    om = MeshTriLite(mesh.p, mesh.t)
    basis = P2Basis(om)
    sigma = shift_estimate(g)
    print(it.arrangement, it.pitch_um, it.wavelength_um, "N", sym.N, "sigma", sigma, flush=True)
    Ke = fe.element_K(hfield.element_matrices(g, basis), g.k0 ** 2, sigma)
    T = fe.FrontTree(sym)
    A, B, *_ = hfield.assemble_hfield_system_fused(g, om)
    A_int, B_int, interior = hfield.restrict_interior(A, B, basis)
    N = sym.N
    ii = np.concatenate([interior, interior + N])
    K = (A_int - sigma * B_int).tocsc()
    rhs = np.zeros(2 * N)
    rhs[ii] = np.random.default_rng(0).standard_normal(len(ii))
    xs = spla.splu(K).solve(rhs[ii])
    print("splu residual", np.linalg.norm(K @ xs - rhs[ii]) / np.linalg.norm(rhs[ii]), flush=True)
    for mode in sys.argv[3:] or ("scalar", "pairs"):
        stats = {"perturbed": 0, "rel": [], "growth": 0.0, "modes": [0, 0, 0, 0], "where": None}
        t0 = time.time()
        Fs, Ds = factor(T, Ke, mode, stats)
        x = solve(T, Fs, Ds, rhs, mode)
        rel = np.sort(np.array(stats["rel"]))
        print(f"{mode:6s}: perturbed {stats['perturbed']}  smallest pivots / block max {rel[:4]}  largest multiplier "
              f"{stats['growth']:.3e}  vs splu {np.linalg.norm(x[ii] - xs) / np.linalg.norm(xs):.3e}  residual "
              f"{np.linalg.norm(K @ x[ii] - rhs[ii]) / np.linalg.norm(rhs[ii]):.3e}  ({time.time() - t0:.0f} s)  pairs by mode "
              f"[2 x 2, scalar order, -, -] {stats['modes']}  largest multiplier at (pair, s2, m, a, b, c) {stats['where']}", flush=True)


if __name__ == "__main__":
    main()
