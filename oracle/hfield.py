"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement of the reference hot path ``solver_fem.py:113-239``
(``TrueVectorialMaxwellSolver.assemble_hfield_system`` / ``solve_vectorial_modes``) on top of the
scikit-fem restatement in ``oracle/p2.py`` and SciPy's own ``eigsh`` (the same third-party code the
reference calls at ``solver_fem.py:197``).

Two assembly shapes are provided:

* ``assemble_hfield_system``        — scikit-fem's loop shape: nine bilinear forms, each evaluated
  for the 36 local (j, i) pairs over ``(ne, 6)`` quadrature arrays with ``epsilon()`` re-evaluated
  inside every kernel call (``solver_fem.py:131-156``), COO->CSR per form, ``bmat``.  This is the
  shape timed as the CPU baseline (``bench.py`` ``cpu_baseline.kind = "port"``).
* ``assemble_hfield_system_fused``  — the same integrals in one einsum pass; used by the parity
  tests where the loop shape would only cost time.  ``tests/test_oracle_hfield.py`` checks the two
  agree to rounding.

PARITY PINNING: see ``oracle/p2.py`` — no reference test pins this path ("parity unpinned" for the
scikit-fem half); closed-form KATs + the reference-generated geometry goldens + SciPy stand in.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import eigsh

from .p2 import MeshTriLite, P2Basis


# ------------------------------------------------------------------------------------------------
# bilinear forms, exactly the integrands of solver_fem.py:131-150  (u = trial j, v = test i)
# ------------------------------------------------------------------------------------------------
def _forms(eps_fn):
    def inv_eps(w):
        return 1.0 / np.real(eps_fn(*w))

    return {
        "kxx": lambda u, gu, v, gv, w: inv_eps(w) * gu[1] * gv[1],          # :132
        "kyy": lambda u, gu, v, gv, w: inv_eps(w) * gu[0] * gv[0],          # :134
        "kxy": lambda u, gu, v, gv, w: -inv_eps(w) * gu[1] * gv[0],         # :136
        "kyx": lambda u, gu, v, gv, w: -inv_eps(w) * gu[0] * gv[1],         # :138
        "div_xx": lambda u, gu, v, gv, w: gu[0] * gv[0],                    # :141
        "div_yy": lambda u, gu, v, gv, w: gu[1] * gv[1],                    # :143
        "div_xy": lambda u, gu, v, gv, w: gu[0] * gv[1],                    # :145
        "mass": lambda u, gu, v, gv, w: u * v,                              # :148
        "mass_eps_inv": lambda u, gu, v, gv, w: inv_eps(w) * u * v,         # :150
    }


def asm(form, basis: P2Basis) -> sp.csr_matrix:
    """scikit-fem ``asm(BilinearForm, basis)``: 36 (j, i) kernels, COO -> CSR (duplicates summed)."""
    ne = basis.mesh.t.shape[1]
    data = np.zeros((6, 6, ne))
    rows = np.zeros((6, 6, ne), dtype=np.int64)
    cols = np.zeros((6, 6, ne), dtype=np.int64)
    w = (basis.qx[0], basis.qx[1])
    for j in range(6):
        for i in range(6):
            integrand = form(basis.phi[j], basis.grad[j], basis.phi[i], basis.grad[i], w)
            data[j, i] = np.sum(integrand * basis.dx, axis=1)
            rows[j, i] = basis.element_dofs[i]
            cols[j, i] = basis.element_dofs[j]
    m = sp.coo_matrix((data.ravel(), (rows.ravel(), cols.ravel())), shape=(basis.N, basis.N)).tocsr()
    m.eliminate_zeros()
    return m


def _block_system(K, k0):
    """solver_fem.py:158-167."""
    alpha_p = 1.0
    k0sq = k0 ** 2
    A_xx = K["kxx"] + alpha_p * K["div_xx"] - k0sq * K["mass"]
    A_yy = K["kyy"] + alpha_p * K["div_yy"] - k0sq * K["mass"]
    A_xy = K["kxy"] + alpha_p * K["div_xy"]
    A_yx = K["kyx"] + alpha_p * K["div_xy"].T
    A = sp.bmat([[A_xx, A_xy], [A_yx, A_yy]], format="csr")
    B = sp.bmat([[K["mass_eps_inv"], None], [None, K["mass_eps_inv"]]], format="csr")
    return A, B


def assemble_hfield_system(geometry, mesh: MeshTriLite):
    """``TrueVectorialMaxwellSolver.assemble_hfield_system`` (solver_fem.py:122-169), skfem loop shape."""
    basis = P2Basis(mesh)
    forms = _forms(geometry.epsilon)
    K = {name: asm(f, basis) for name, f in forms.items()}
    A, B = _block_system(K, geometry.k0)
    return A, B, basis, K["div_xx"], K["div_yy"], K["div_xy"], K["mass_eps_inv"]


def element_matrices(geometry, basis: P2Basis):
    """All nine 6x6 element matrices at once: dict name -> (ne, 6[i=test/row], 6[j=trial/col])."""
    qx, qy = basis.qx
    inv_eps = 1.0 / np.real(geometry.epsilon(qx, qy))                        # (ne, 6)
    w1 = basis.dx
    we = basis.dx * inv_eps
    gx = basis.grad[:, 0]                                                     # (6, ne, 6)
    gy = basis.grad[:, 1]
    phi = basis.phi

    def bil(a_trial, b_test, w):                                             # out[e, i, j]
        return np.einsum("jeq,ieq,eq->eij", a_trial, b_test, w, optimize=True)

    return {
        "kxx": bil(gy, gy, we), "kyy": bil(gx, gx, we),
        "kxy": -bil(gy, gx, we), "kyx": -bil(gx, gy, we),
        "div_xx": bil(gx, gx, w1), "div_yy": bil(gy, gy, w1), "div_xy": bil(gx, gy, w1),
        "mass": bil(phi, phi, w1), "mass_eps_inv": bil(phi, phi, we),
    }


def assemble_hfield_system_fused(geometry, mesh: MeshTriLite, eliminate_zeros: bool = True):
    """Same matrices as :func:`assemble_hfield_system`, one pass over the elements."""
    basis = P2Basis(mesh)
    em = element_matrices(geometry, basis)
    ed = basis.element_dofs
    rows = np.broadcast_to(ed.T[:, :, None], (ed.shape[1], 6, 6)).ravel()
    cols = np.broadcast_to(ed.T[:, None, :], (ed.shape[1], 6, 6)).ravel()
    K = {}
    for name, d in em.items():
        m = sp.coo_matrix((d.ravel(), (rows, cols)), shape=(basis.N, basis.N)).tocsr()
        if eliminate_zeros:
            m.eliminate_zeros()
        K[name] = m
    A, B = _block_system(K, geometry.k0)
    return A, B, basis, K["div_xx"], K["div_yy"], K["div_xy"], K["mass_eps_inv"]


# ------------------------------------------------------------------------------------------------
def shift_estimate(geometry) -> float:
    """LP01 b-V estimate of the shift, solver_fem.py:187-193."""
    n_core, n_clad = geometry.n_core, geometry.n_clad
    NA = np.sqrt(max(n_core ** 2 - n_clad ** 2, 1e-6))
    r_mean = np.mean(geometry.core_radii)
    V_geom = geometry.k0 * r_mean * NA
    b_approx = max((1.0 - 2.405 / max(V_geom, 2.41)) ** 2, 0.05)
    n_eff_est = np.sqrt(n_clad ** 2 + b_approx * (n_core ** 2 - n_clad ** 2))
    return float((geometry.k0 * float(np.clip(n_eff_est, n_clad + 0.05, n_core - 0.005))) ** 2)


def polarization_from_interp(evec_x, evec_y, x_dof, y_dof, geometry):
    """solver_fem.py:68-107."""
    in_core = np.zeros(len(x_dof), dtype=bool)
    for (cx, cy), r in zip(geometry.positions, geometry.core_radii):
        in_core |= (x_dof - cx) ** 2 + (y_dof - cy) ** 2 <= r ** 2
    mask = in_core if np.any(in_core) else np.ones(len(x_dof), dtype=bool)
    P_x = float(np.sum(evec_x[mask] ** 2)) + 1e-30
    P_y = float(np.sum(evec_y[mask] ** 2)) + 1e-30
    ratio = P_x / P_y
    PDL = float(np.clip(10.0 * np.log10(max(P_x, P_y) / min(P_x, P_y)), 0.0, 50.0))
    if ratio > 10.0:
        pol = "TE-like"
    elif ratio > 2.5:
        pol = "HE-like"
    elif ratio > 0.4:
        pol = "Hybrid"
    elif ratio > 0.1:
        pol = "EH-like"
    else:
        pol = "TM-like"
    return pol, PDL, P_x, P_y


def restrict_interior(A, B, basis):
    """solver_fem.py:179-184."""
    N = basis.N
    boundary_dofs = basis.get_dofs().all()
    interior = np.setdiff1d(np.arange(N), boundary_dofs)
    idx = np.concatenate([interior, interior + N])
    return A[idx, :][:, idx], B[idx, :][:, idx], interior


def postprocess_modes(geometry, beta_sq, evecs, interior, basis, Dxx, Dyy, Dxy, return_all=False):
    """solver_fem.py:199-239 (per-mode quantities, divergence / radiation filters, ordering)."""
    k0 = geometry.k0
    n_core, n_clad = geometry.n_core, geometry.n_clad
    N_solve = len(interior)
    x_int = basis.doflocs[0][interior]
    y_int = basis.doflocs[1][interior]
    in_core = np.zeros(N_solve, dtype=bool)
    for (cx, cy), r in zip(geometry.positions, geometry.core_radii):
        in_core |= (x_int - cx) ** 2 + (y_int - cy) ** 2 <= r ** 2
    frac_core = np.sum(in_core) / N_solve
    # the reference re-extracts these three per mode (:214); identical values, extracted once here
    Dxx_i = Dxx[interior, :][:, interior]
    Dxy_i = Dxy[interior, :][:, interior]
    Dyy_i = Dyy[interior, :][:, interior]
    raw = []
    for i in range(len(beta_sq)):
        b2 = beta_sq[i]
        if b2 <= 0:
            continue
        beta = np.sqrt(b2)
        ne = beta / k0
        if ne <= n_clad or ne >= n_core * 1.01:
            continue
        vx = evecs[:N_solve, i].copy()
        vy = evecs[N_solve:, i].copy()
        nrm = np.sqrt(np.sum(vx ** 2) + np.sum(vy ** 2)) + 1e-30
        vx /= nrm
        vy /= nrm
        div_energy = float(vx @ (Dxx_i @ vx) + 2 * vx @ (Dxy_i @ vy) + vy @ (Dyy_i @ vy))
        div_ratio = div_energy / max(b2, 1e-12)
        e2 = vx ** 2 + vy ** 2
        conf = float(np.sum(e2[in_core]) / np.sum(e2))
        pol, PDL, P_x, P_y = polarization_from_interp(vx, vy, x_int, y_int, geometry)
        raw.append({"n_eff": float(ne), "beta": float(beta), "Ex_dofs": vx, "Ey_dofs": vy,
                    "P_x": P_x, "P_y": P_y, "PDL_dB": PDL, "polarization": pol,
                    "confinement": conf, "core_overlap": conf, "div_ratio": div_ratio,
                    "is_vectorial": True, "method": "H-field_V18.10"})
    if not raw:
        return []
    dr = np.array([m["div_ratio"] for m in raw])
    thr = max(np.median(dr) * 10, dr.min() * 50, 1e-6)
    phys = [m for m in raw if m["div_ratio"] <= thr]
    conf_thr = max(5.0 * frac_core, 0.05)
    guided = [m for m in phys if m["confinement"] >= conf_thr]
    if not guided:
        guided = phys
    guided.sort(key=lambda m: m["n_eff"], reverse=True)
    return guided


def solve_vectorial_modes(geometry, mesh: MeshTriLite, n_modes_target: int = 20, fused: bool = False,
                          tol: float = 1e-7, return_raw: bool = False, timings: dict | None = None):
    """``TrueVectorialMaxwellSolver.solve_vectorial_modes`` (solver_fem.py:171-239)."""
    import time
    t0 = time.perf_counter()
    asm_fn = assemble_hfield_system_fused if fused else assemble_hfield_system
    A, B, basis, Dxx, Dyy, Dxy, M_inv = asm_fn(geometry, mesh)
    t1 = time.perf_counter()
    A_int, B_int, interior = restrict_interior(A, B, basis)
    N_solve = len(interior)
    sigma = shift_estimate(geometry)
    n_req = min(n_modes_target + 12, 2 * N_solve - 4)
    t2 = time.perf_counter()
    beta_sq, evecs = eigsh(A_int, k=n_req, M=B_int, sigma=sigma, which="LM", tol=tol, maxiter=12000)
    t3 = time.perf_counter()
    modes = postprocess_modes(geometry, beta_sq, evecs, interior, basis, Dxx, Dyy, Dxy)
    t4 = time.perf_counter()
    if timings is not None:
        timings.update(assembly=t1 - t0, restrict=t2 - t1, eigsh=t3 - t2, post=t4 - t3, total=t4 - t0,
                       N=basis.N, N_solve=N_solve, n=2 * N_solve, nnzA=A_int.nnz, nnzB=B_int.nnz)
    if return_raw:
        return modes, dict(beta_sq=beta_sq, evecs=evecs, sigma=sigma, interior=interior, basis=basis,
                           A_int=A_int, B_int=B_int)
    return modes
