"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement of the reference's scalar P2 solver ``ScalarHelmholtzSolver.solve``
(``solver_fem.py:245-276``) on the scikit-fem restatement of ``oracle/p2.py`` and SciPy's own
``eigsh``:  (K - k0^2 M_eps) u = lambda M u  on ALL P2 DOFs (natural boundary, no Dirichlet elimination), lambda = -beta^2,
shift sigma = -(k0 (n_core - 0.008))^2, ``k = min(n_modes_target + 8, N - 4)``, ``tol = 1e-6``, ``maxiter = 6000``.

PARITY PINNING: as for ``oracle/hfield.py`` — the reference ships no test or golden vector for this solver and
scikit-fem cannot run here ("parity unpinned" for the scikit-fem half); the quadrature / numbering restatement is the
one pinned by the closed-form KATs of ``tests/test_oracle_p2.py``, the eigen-solve is SciPy's.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import eigsh

from .hfield import asm
from .p2 import MeshTriLite, P2Basis


def scalar_forms(eps_fn):
    """The three integrands of ``solver_fem.py:252-257`` (u = trial, v = test)."""
    return {
        "stiff": lambda u, gu, v, gv, w: gu[0] * gv[0] + gu[1] * gv[1],                 # dot(grad(u), grad(v))
        "mass": lambda u, gu, v, gv, w: u * v,
        "eps_m": lambda u, gu, v, gv, w: np.real(eps_fn(*w)) * u * v,
    }


def element_matrices(geometry, basis: P2Basis):
    """dict name -> (ne, 6[test i], 6[trial j]) in one pass (same integrals as the three ``asm`` calls)."""
    qx, qy = basis.qx
    eps = np.real(geometry.epsilon(qx, qy))
    w1 = basis.dx
    gx, gy, phi = basis.grad[:, 0], basis.grad[:, 1], basis.phi

    def bil(a_trial, b_test, w):
        return np.einsum("jeq,ieq,eq->eij", a_trial, b_test, w, optimize=True)

    return {"stiff": bil(gx, gx, w1) + bil(gy, gy, w1), "mass": bil(phi, phi, w1), "eps_m": bil(phi, phi, w1 * eps)}


def assemble(geometry, mesh: MeshTriLite, fused: bool = True, eliminate_zeros: bool = True):
    """K, M, M_eps (CSR N x N) and the basis, ``solver_fem.py:251-259``."""
    basis = P2Basis(mesh)
    if not fused:
        forms = scalar_forms(geometry.epsilon)
        return asm(forms["stiff"], basis), asm(forms["mass"], basis), asm(forms["eps_m"], basis), basis
    em = element_matrices(geometry, basis)
    ed = basis.element_dofs
    rows = np.broadcast_to(ed.T[:, :, None], (ed.shape[1], 6, 6)).ravel()
    cols = np.broadcast_to(ed.T[:, None, :], (ed.shape[1], 6, 6)).ravel()
    out = []
    for name in ("stiff", "mass", "eps_m"):
        m = sp.coo_matrix((em[name].ravel(), (rows, cols)), shape=(basis.N, basis.N)).tocsr()
        if eliminate_zeros:
            m.eliminate_zeros()
        out.append(m)
    return out[0], out[1], out[2], basis


def shift(geometry) -> float:
    """``solver_fem.py:260``."""
    return float(-(geometry.k0 * (geometry.n_core - 0.008)) ** 2)


def solve(geometry, mesh: MeshTriLite, n_modes_target: int = 20, fused: bool = True, tol: float = 1e-6,
          return_raw: bool = False):
    """``ScalarHelmholtzSolver.solve`` (``solver_fem.py:250-276``): list of mode dicts, n_eff descending."""
    K, M, Me, basis = assemble(geometry, mesh, fused=fused)
    k0 = geometry.k0
    sigma = shift(geometry)
    k = min(n_modes_target + 8, basis.N - 4)
    A = (K - k0 ** 2 * Me).tocsr()
    evals, evecs = eigsh(A, k=k, M=M, sigma=sigma, which="LM", tol=tol, maxiter=6000)
    x_dof, y_dof = basis.doflocs
    in_core = np.zeros(len(x_dof), dtype=bool)
    for (cx, cy), r in zip(geometry.positions, geometry.core_radii):
        in_core |= (x_dof - cx) ** 2 + (y_dof - cy) ** 2 <= r ** 2
    modes = []
    for i in range(len(evals)):
        if evals[i] >= 0:
            continue
        ne = np.sqrt(-evals[i]) / k0
        if ne <= geometry.n_clad or ne >= geometry.n_core * 1.005:
            continue
        v = evecs[:, i].copy()
        v /= np.sqrt(float(v @ (M @ v))) + 1e-30
        conf = float(np.sum(v[in_core] ** 2) / np.sum(v ** 2))
        modes.append({"n_eff": float(ne), "beta": float(k0 * ne), "field_vector": v, "confinement": conf,
                      "core_overlap": conf, "PDL_dB": 0.0, "polarization": "scalar", "is_vectorial": False})
    modes.sort(key=lambda m: m["n_eff"], reverse=True)
    if return_raw:
        return modes, dict(evals=evals, evecs=evecs, sigma=sigma, A=A, M=M, basis=basis)
    return modes
