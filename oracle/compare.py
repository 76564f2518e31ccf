"""Test infrastructure (checker only, never on the product path): compares two lists of mode records.

The reference fixes neither the sign of an eigenvector nor the basis inside a (near-)degenerate
eigenspace -- ``eigsh`` returns whatever ARPACK's start vector produced (``solver_fem.py:197``) -- so the
field comparison is sign-invariant for isolated modes and a subspace distance for clusters of modes whose
``n_eff`` differ by less than ``rel_gap`` (the hexagonal symmetry of the 7/19-core sections gives exact
pairs in the continuum that an unsymmetric mesh splits by 1e-10..1e-7).
"""
from __future__ import annotations

import numpy as np


def stack_fields(mode) -> np.ndarray:
    return np.concatenate([np.asarray(mode["Ex_dofs"]), np.asarray(mode["Ey_dofs"])])


def column_errors(V: np.ndarray, U: np.ndarray, keys: np.ndarray, rel_gap: float) -> np.ndarray:
    """V, U: (n, k) columns ordered like ``keys`` (monotone).  Per-column L2 distance of the normalised
    columns up to sign; for a cluster (consecutive |Δkey| < rel_gap |key|) the spectral norm of
    (I - Qu Qu^T) Qv, assigned to every member."""
    k = len(keys)
    errs = np.zeros(k)
    i = 0
    while i < k:
        j = i + 1
        while j < k and abs(keys[j] - keys[j - 1]) < rel_gap * abs(keys[j]):
            j += 1
        a = V[:, i:j] / np.linalg.norm(V[:, i:j], axis=0)
        b = U[:, i:j] / np.linalg.norm(U[:, i:j], axis=0)
        if j - i == 1:
            errs[i] = min(np.linalg.norm(a[:, 0] - b[:, 0]), np.linalg.norm(a[:, 0] + b[:, 0]))
        else:
            Qa, _ = np.linalg.qr(a)
            Qb, _ = np.linalg.qr(b)
            errs[i:j] = np.linalg.norm(Qa - Qb @ (Qb.T @ Qa), 2)
        i = j
    return errs


def mode_field_errors(modes, ref, rel_gap: float = 1e-6) -> np.ndarray:
    """Field error per mode between two equally long, equally ordered lists of mode records."""
    if len(modes) != len(ref):
        raise ValueError(f"{len(modes)} modes vs {len(ref)} reference modes")
    if not modes:
        return np.zeros(0)
    V = np.stack([stack_fields(m) for m in modes], axis=1)
    U = np.stack([stack_fields(m) for m in ref], axis=1)
    keys = np.array([m["n_eff"] for m in ref], dtype=float)
    return column_errors(V, U, keys, rel_gap)
