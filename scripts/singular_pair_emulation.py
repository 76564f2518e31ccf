"""CPU only: the one case where this path and the reference's SuperLU differ in KIND.  The
multifrontal LDL^T pivots inside node pairs without permutation; SuperLU pivots across rows.  Choose sigma so that the
FIRST pair a leaf front eliminates -- E = (A - sigma B) restricted to the (Hx, Hy) DOFs of that node, nothing eliminated
before it -- is singular as a whole: K itself is perfectly regular (sigma is no eigenvalue of the pencil), SuperLU does
not even notice, but an LDL^T in this order meets a vanishing pivot.  The emulation (the pair rule of kernels_front.hip,
scripts/pair_pivot_emulation.py) shows what static perturbation + refinement make of it, for several replacement values:

    singular_pair_emulation.py [refinement] [replacement exponents ...]     (default 0.3; -13 -10 -8)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import scipy.sparse.linalg as spla

import front_emulation as fe
import pair_pivot_emulation as ppe
from oracle import hfield
from oracle.p2 import MeshTriLite, P2Basis
from pl_fem_vectoriel_amd import MCFGeometry, _native
from pl_fem_vectoriel_amd.mesh import generate_mesh

NB = 32


def ldl_pairs(Fm, s2, stats, repl):
    """ppe.ldl_pairs with the replacement value of a vanishing pivot as a parameter (detection stays at 1e-13 of the
    pair's rows) and the vanishing-determinant guard of the 2 x 2 form."""
    F = Fm.copy()
    Dinv = np.zeros((s2, 2))
    for k in range(0, s2, 2):
        if k % NB == 0:
            blk = F[k:min(k + NB, s2), k:min(k + NB, s2)]
            rowmax = np.abs(blk).max(axis=1)
        a, b, c = F[k, k], F[k + 1, k], F[k + 1, k + 1]
        rm = max(rowmax[k % NB], rowmax[k % NB + 1])
        thr, rep = max(1e-13 * rm, 1e-300), max(repl * rm, 1e-300)
        det = a * c - b * b
        s = max(abs(a), abs(b), abs(c))
        if b * b * abs(det) <= a * a * s * s:
            for j in (k, k + 1):
                d = F[j, j]
                if not abs(d) >= thr:
                    d = -rep if d < 0 else rep
                    stats["perturbed"] += 1
                l = F[j + 1:, j] / d
                stats["growth"] = max(stats["growth"], float(np.abs(l).max()) if len(l) else 0.0)
                F[j + 1:, j + 1:] -= np.outer(l, F[j + 1:, j])
                F[j + 1:, j] = l
                Dinv[j] = (1.0 / d, 0.0)
        else:
            if not abs(det) >= thr * s:
                det = -rep * s if det < 0 else rep * s
                stats["perturbed"] += 1
            e11, e12, e22 = c / det, -b / det, a / det
            Dinv[k] = (e11, e12)
            Dinv[k + 1] = (e22, e12)
            C = F[k + 2:, k:k + 2].copy()
            Lc = np.stack([C[:, 0] * e11 + C[:, 1] * e12, C[:, 0] * e12 + C[:, 1] * e22], 1)
            stats["growth"] = max(stats["growth"], float(np.abs(Lc).max()) if len(Lc) else 0.0)
            F[k + 2:, k + 2:] -= Lc @ C.T
            F[k + 2:, k:k + 2] = Lc
            F[k + 1, k] = 0.0
    return F, Dinv


def factor(T, Ke, stats, repl):
    Fs, Ds, S = [None] * T.nf, [None] * T.nf, [None] * T.nf
    for f in range(T.nf - 1, -1, -1):
        Fm = fe.assemble_front(T, f, Ke, S)
        s2 = T.s2(f)
        F, D = ldl_pairs(Fm, s2, stats, repl)
        Fs[f], Ds[f] = ppe.finish(F, s2), D
        S[f] = Fs[f][s2:, s2:]
    return Fs, Ds


def singular_pair_sigma(sym, A, B, which=0):
    """sigma at which the first pair of the first leaf front with owned nodes is singular: an eigenvalue of the 2 x 2 pencil
    (A_pp, B_pp) of that node's (Hx, Hy) DOFs.  Returns (sigma, node)."""
    N = sym.N
    fs_true = sym.array("fs_true")
    fptr = sym.array("fnode_ptr")
    fnodes = sym.array("fnodes")
    nf = len(fs_true)
    leaf0 = (nf + 1) // 2 - 1
    f = next(q for q in range(leaf0, nf) if fs_true[q] > 0)
    node = int(fnodes[fptr[f]])
    idx = [node, N + node]
    App = A[idx][:, idx].toarray()
    Bpp = B[idx][:, idx].toarray()
    w = np.linalg.eigvals(np.linalg.solve(Bpp, App))
    w = np.sort(w.real)
    return float(w[which]), node, f


def main():
    refinement = float(sys.argv[1]) if len(sys.argv) > 1 else 0.3
    exps = [float(x) for x in sys.argv[2:]] or [-13.0, -10.0, -8.0]
    g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    mesh = generate_mesh(g, refinement, 0)
    sym = _native.Symbolic(mesh.p, mesh.t)
    om = MeshTriLite(mesh.p, mesh.t)
    basis = P2Basis(om)
    A, B, *_ = hfield.assemble_hfield_system_fused(g, om)
    A_int, B_int, interior = hfield.restrict_interior(A, B, basis)
    N = sym.N
    ii = np.concatenate([interior, interior + N])
    sigma, node, front = singular_pair_sigma(sym, A.tocsr(), B.tocsr())
    print(f"N {N}  front {front}  node {node}  sigma* {sigma!r}", flush=True)
    K = (A_int - sigma * B_int).tocsc()
    lu = spla.splu(K)
    rhs = np.zeros(2 * N)
    rhs[ii] = np.random.default_rng(0).standard_normal(len(ii))
    xs = lu.solve(rhs[ii])
    print("splu residual", np.linalg.norm(K @ xs - rhs[ii]) / np.linalg.norm(rhs[ii]), " |x|", np.linalg.norm(xs), flush=True)
    Ke = fe.element_K(hfield.element_matrices(g, basis), g.k0 ** 2, sigma)
    T = fe.FrontTree(sym)
    for e in exps:
        stats = {"perturbed": 0, "growth": 0.0}
        Fs, Ds = factor(T, Ke, stats, 10.0 ** e)
        x = ppe.solve(T, Fs, Ds, rhs, "pairs")
        line = f"replacement 1e{e:+.0f} x rowmax: perturbed {stats['perturbed']} largest multiplier {stats['growth']:.2e}  vs splu"
        for it in range(4):
            line += f"  {np.linalg.norm(x[ii] - xs) / np.linalg.norm(xs):.2e}"
            r = np.zeros(2 * N)
            r[ii] = rhs[ii] - K @ x[ii]
            x = x + ppe.solve(T, Fs, Ds, r, "pairs")
        print(line + "   (0, 1, 2, 3 refinement passes)", flush=True)


if __name__ == "__main__":
    main()
