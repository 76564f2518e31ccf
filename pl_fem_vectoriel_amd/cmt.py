"""Rigorous coupled-mode coupling matrix on the GPU.

Reference: ``CoupledModeTheory._compute_rigorous_coupling`` (``config.py:274-322``; the file is misnamed, it holds the
CMT module): for the scalar mode records of ``ScalarHelmholtzSolver`` (``field_vector`` over all P2 DOFs),

    H_ii = beta_i,   H_ij = H_ji = (omega / 4) E_i^H M_deps E_j / sqrt(P_i P_j + 1e-15)  (i < j),   P = E^H E,
    M_deps = asm((eps - mean(eps)) u v).

The integrals run in ``libplfem_hip.so`` (``plfem_cmt_coupling``: element-batch weighted mass, ordered CSR gather, one
SpMV + one panel product per mode); the scaling and the diagonal stay here, as in the reference.  Only the coupling
matrix is built: the propagation half of the reference class (``expm`` / RK45 on an n_modes x n_modes system,
``config.py:60-250``) is dense small-matrix host work outside the hot path.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import numpy as np

from . import _native
from .solver_fem import _core_table


class CoupledModeTheory:
    """``CoupledModeTheory(omega, coupling_method)`` with the reference's constructor checks (``config.py:45-55``)."""

    def __init__(self, omega: float, coupling_method: str = "approximate", device: Optional[int] = None):
        self.omega = omega
        self.coupling_method = coupling_method
        if coupling_method not in ["approximate", "rigorous"]:
            raise ValueError("coupling_method doit être 'approximate' ou 'rigorous'")
        self.device = device
        self.last_stats: Dict = {}

    def _compute_rigorous_coupling(self, modes_i: List[Dict], modes_j: List[Dict], geometry, basis) -> np.ndarray:
        """``basis``: the mesh the fields live on (a ``TriMesh``, or any object with ``.mesh`` or ``.p`` / ``.t`` — the
        reference passes the scikit-fem ``Basis`` of that mesh)."""
        import torch

        mesh = getattr(basis, "mesh", basis)
        n = len(modes_i)
        if len(modes_j) != n:
            raise ValueError("modes_i and modes_j must have the same length")
        H = np.zeros((n, n), dtype=complex)
        for i in range(n):
            H[i, i] = modes_i[i]["beta"]
        if n == 0:
            return H
        Ei = np.ascontiguousarray(np.array([m["field_vector"] for m in modes_i]))
        Ej = np.ascontiguousarray(np.array([m["field_vector"] for m in modes_j]))
        if np.iscomplexobj(Ei) or np.iscomplexobj(Ej):
            raise NotImplementedError("complex fields: the scalar solver of the reference returns real vectors")
        sym = _native.Symbolic(mesh.p, mesh.t, dofs_per_node=1, dirichlet=False)
        if Ei.shape != (n, sym.N):
            raise ValueError(f"field_vector must have one entry per P2 DOF of the mesh ({sym.N})")
        ctx = _native.Context(sym, self.device, max_ncv=max(65, n))
        try:
            di = torch.from_numpy(Ei.astype(np.float64)).to(ctx.tdev)
            dj = torch.from_numpy(Ej.astype(np.float64)).to(ctx.tdev)
            raw, Pi, Pj, mean = ctx.cmt_coupling(di, dj, _core_table(geometry), geometry.n_core ** 2, geometry.n_clad ** 2)
        finally:
            ctx.close()
        for i in range(n):
            for j in range(i + 1, n):
                C = raw[i, j] / np.sqrt(Pi[i] * Pj[j] + 1e-15)
                H[i, j] = H[j, i] = C * self.omega / 4.0
        self.last_stats = {"eps_mean": mean, "N": sym.N}
        return H
