"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement of the reference's rigorous coupled-mode coupling matrix ``CoupledModeTheory._compute_rigorous_coupling``
(``config.py:274-322``) on the scikit-fem restatement of ``oracle/p2.py``:

    H_ii = beta_i,      H_ij = H_ji = (omega / 4) E_i^H M_deps E_j / sqrt(P_i P_j + 1e-15)   (i < j),   P = E^H E,
    M_deps = asm( (eps - mean(eps)) u v )   with eps = geometry.epsilon at the quadrature points and mean() the plain
             (unweighted) mean over ALL quadrature points of the mesh (config.py:297-300: np.mean of the (ne, 6) array).

scikit-fem's ``BilinearForm`` assembles into float64 (its default dtype), so the imaginary part the PML gives
``geometry.epsilon`` is discarded at assembly, exactly as on the vectorial path: the weight is
Re(eps) - mean(Re(eps)).  H is returned complex like the reference's (``np.zeros((n, n), dtype=complex)``).

PARITY PINNING: ``config.py`` is not importable (relative imports of absent modules) and holds no test for
this function: "parity unpinned", closed-form checks in tests/ instead (constant eps -> zero coupling; H = H^T; scaling).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .p2 import P2Basis


def delta_eps_mass(geometry, basis: P2Basis):
    """``asm(epsilon_product, basis)`` of ``config.py:296-303`` (CSR N x N) and the mean permittivity it subtracts."""
    qx, qy = basis.qx
    eps = np.real(geometry.epsilon(qx, qy))                  # (ne, 6)
    mean = float(np.mean(eps))
    w = basis.dx * (eps - mean)
    em = np.einsum("jeq,ieq,eq->eij", basis.phi, basis.phi, w, optimize=True)
    ed = basis.element_dofs
    rows = np.broadcast_to(ed.T[:, :, None], (ed.shape[1], 6, 6)).ravel()
    cols = np.broadcast_to(ed.T[:, None, :], (ed.shape[1], 6, 6)).ravel()
    return sp.coo_matrix((em.ravel(), (rows, cols)), shape=(basis.N, basis.N)).tocsr(), mean


def rigorous_coupling(modes_i, modes_j, geometry, basis: P2Basis, omega: float) -> np.ndarray:
    """``_compute_rigorous_coupling`` (``config.py:274-322``)."""
    n = len(modes_i)
    H = np.zeros((n, n), dtype=complex)
    for i in range(n):
        H[i, i] = modes_i[i]["beta"]
    M_eps, _ = delta_eps_mass(geometry, basis)
    for i in range(n):
        E_i = modes_i[i]["field_vector"]
        P_i = np.real(E_i.conj() @ E_i)
        for j in range(i + 1, n):
            E_j = modes_j[j]["field_vector"]
            P_j = np.real(E_j.conj() @ E_j)
            C = E_i.conj() @ (M_eps @ E_j)
            C /= np.sqrt(P_i * P_j + 1e-15)
            C *= omega / 4.0
            H[i, j] = H[j, i] = C
    return H
