"""NumPy emulation of the multifrontal block-LDL^T factorisation / solve that libplfem_hip.so runs,
driven by the same symbolic arrays (test infrastructure: checks the front tree on the CPU and the
HIP kernels front by front on the GPU)."""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla


class FrontTree:
    def __init__(self, sym):
        self.sym = sym
        self.N = sym.N
        self.ne = sym.ne
        self.fs = sym.array("fs")
        self.fb = sym.array("fb")
        self.fptr = sym.array("fnode_ptr")
        self.fnodes = sym.array("fnodes")
        self.c0 = sym.array("cinv0")
        self.c1 = sym.array("cinv1")
        self.epos = sym.array("epos").reshape(6, sym.ne)
        self.lptr = sym.array("leaf_elem_ptr")
        self.lel = sym.array("leaf_elems")
        self.foff = sym.array("foff")
        self.L = sym.info["levels"]
        self.nf = sym.info["nfronts"]
        self.leaf0 = (1 << self.L) - 1

    def m(self, f):
        return 2 * int(self.fs[f] + self.fb[f])

    def s2(self, f):
        return 2 * int(self.fs[f])

    def nodes(self, f):
        return self.fnodes[self.fptr[f]:self.fptr[f] + self.fs[f] + self.fb[f]]

    def device_front(self, ctx, f, with_schur=False):
        """Front f as the device context stores it, rebuilt as one m x m array [row, col] (symbolic.h): [F11; F21] (m x s2,
        leading dimension m) and Z^T (s2 x b2, leading dimension s2) from the permanent storage; the Schur complement F22
        (b2 x b2) from the arena of the front's tree level -- only meaningful while that arena has not been reused (after a
        complete factorisation: levels 0 and 1), NaN otherwise."""
        m, s2 = self.m(f), self.s2(f)
        b2 = m - s2
        off = int(self.foff[f])
        F = np.full((m, m), np.nan)
        if s2:
            F[:, :s2] = ctx.debug_copy("front", off, m * s2).reshape(s2, m).T
            if b2:
                F[:s2, s2:] = ctx.debug_copy("front", off + m * s2, s2 * b2).reshape(b2, s2).T
        if with_schur and b2:
            level = int(np.floor(np.log2(f + 1)))
            arena = (int(self.sym.info["arena_doubles"]) + 31) & ~31
            soff = int(self.sym.array("soff")[f])
            F[s2:, s2:] = ctx.debug_copy("schur", (level & 1) * arena + soff, b2 * b2).reshape(b2, b2).T
        return F


def element_K(em, k0sq, sigma):
    """12x12 element matrices of K = A - sigma B in the interleaved (node, component) DOF order."""
    Axx = em["kxx"] + em["div_xx"] - k0sq * em["mass"]
    Ayy = em["kyy"] + em["div_yy"] - k0sq * em["mass"]
    Axy = em["kxy"] + em["div_xy"]
    Ayx = em["kyx"] + np.transpose(em["div_xy"], (0, 2, 1))
    Mi = em["mass_eps_inv"]
    ne = Axx.shape[0]
    Ke = np.zeros((ne, 12, 12))
    Ke[:, 0::2, 0::2] = Axx - sigma * Mi
    Ke[:, 0::2, 1::2] = Axy
    Ke[:, 1::2, 0::2] = Ayx
    Ke[:, 1::2, 1::2] = Ayy - sigma * Mi
    return Ke


def assemble_front(T: FrontTree, f, Ke, S):
    """Front f before elimination: leaf = its elements, internal = extend-add of the children's S."""
    mn = int(T.fs[f] + T.fb[f])
    m = 2 * mn
    fn = T.nodes(f)
    Fm = np.zeros((m, m))
    pad = np.nonzero(fn < 0)[0]
    Fm[2 * pad, 2 * pad] = 1.0
    Fm[2 * pad + 1, 2 * pad + 1] = 1.0
    if f >= T.leaf0:
        lf = f - T.leaf0
        for e in T.lel[T.lptr[lf]:T.lptr[lf + 1]]:
            pos = T.epos[:, e]
            dofs = np.stack([2 * pos, 2 * pos + 1], 1).ravel()
            ok = np.repeat(pos >= 0, 2)
            ii = dofs[ok]
            Fm[np.ix_(ii, ii)] += Ke[e][np.ix_(ok, ok)]
    else:
        for ch, ci in ((2 * f + 1, T.c0), (2 * f + 2, T.c1)):
            inv = ci[T.fptr[f]:T.fptr[f] + mn]
            s2c = T.s2(ch)
            ok = inv >= 0
            pidx = np.nonzero(ok)[0]
            pd = np.stack([2 * pidx, 2 * pidx + 1], 1).ravel()
            cd = np.stack([s2c + 2 * inv[ok], s2c + 2 * inv[ok] + 1], 1).ravel()
            Fm[np.ix_(pd, pd)] += S[ch][np.ix_(cd - s2c, cd - s2c)]
    return Fm


def ldl_partial(Fm, s2):
    """Partial block LDL^T of the first s2 pivots, node pair by node pair in the static order (local DOFs 2q, 2q+1, no
    permutation; kernels_front.hip): two scalar pivots (a, then c - b^2 / a) or one 2 x 2 pivot, whichever amplifies rounding
    errors less ((b / a)^2 against max|E|^2 / |det|).  Returns the storage the HIP path leaves in F: lower(F11) = L11^-1,
    upper(F11) = L11^-T, F21 = Z = L21 L11^-1, F12 = Z^T, F22 = S; and D^-1 as (diagonal, off-diagonal) per row."""
    F = Fm.copy()
    Dinv = np.zeros((s2, 2))
    for k in range(0, s2, 2):
        a, b, c = F[k, k], F[k + 1, k], F[k + 1, k + 1]
        det = a * c - b * b
        s = max(abs(a), abs(b), abs(c))
        if b * b * abs(det) <= a * a * s * s:
            for j in (k, k + 1):                             # two steps of the scalar LDL^T
                d = F[j, j]
                l = F[j + 1:, j] / d
                F[j + 1:, j + 1:] -= np.outer(l, F[j + 1:, j])
                F[j + 1:, j] = l
                Dinv[j] = (1.0 / d, 0.0)
        else:
            e11, e12, e22 = c / det, -b / det, a / det
            Dinv[k] = (e11, e12)
            Dinv[k + 1] = (e22, e12)
            C = F[k + 2:, k:k + 2].copy()
            Lc = np.stack([C[:, 0] * e11 + C[:, 1] * e12, C[:, 0] * e12 + C[:, 1] * e22], 1)
            F[k + 2:, k + 2:] -= Lc @ C.T
            F[k + 2:, k:k + 2] = Lc
            F[k + 1, k] = 0.0
    L11 = np.tril(F[:s2, :s2], -1) + np.eye(s2)
    X = sla.solve_triangular(L11, np.eye(s2), lower=True, unit_diagonal=True) if s2 else np.zeros((0, 0))
    out = F.copy()
    out[:s2, :s2] = np.tril(X) + np.tril(X, -1).T
    Z = F[s2:, :s2] @ np.tril(X)
    out[s2:, :s2] = Z
    out[:s2, s2:] = Z.T
    return out, Dinv


def factor(T: FrontTree, Ke):
    Fs = [None] * T.nf
    Ds = [None] * T.nf
    S = [None] * T.nf
    for f in range(T.nf - 1, -1, -1):
        Fm = assemble_front(T, f, Ke, S)
        s2 = T.s2(f)
        Fs[f], Ds[f] = ldl_partial(Fm, s2)
        S[f] = Fs[f][s2:, s2:]
    return Fs, Ds


def solve(T: FrontTree, Fs, Ds, rhs):
    """Forward / backward sweeps exactly as the HIP kernels do them.  rhs, result: 2N-vectors."""
    N = T.N
    W = [None] * T.nf
    Y = [None] * T.nf
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
        t = np.array([F[:i + 1, i] @ r[:i + 1] for i in range(s2)]) if s2 else np.zeros(0)
        ys = Ds[f][:, 0] * t + Ds[f][:, 1] * t.reshape(-1, 2)[:, ::-1].ravel()        # D^-1 t, partner of row i = i ^ 1
        w[s2:] -= F[s2:, :s2] @ r                                  # u = w_b - Z r
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
        v = np.concatenate([Y[f], -xb])                            # [ys ; -x_b]
        xo = np.array([F[j:, j] @ v[j:] for j in range(s2)])      # L11^-T ys - Z^T x_b
        ok = node[:s2] >= 0
        x[comp[:s2][ok] * N + node[:s2][ok]] = xo[ok]
    return x
