"""Oracle self-consistency: skfem loop shape vs fused pass, pencil identities, SciPy eigsh semantics."""
import numpy as np
import scipy.sparse.linalg as spla

from oracle import hfield
from oracle.p2 import MeshTriLite
from pl_fem_vectoriel_amd.mesh import generate_mesh


def test_loop_shape_equals_fused_and_pencil_identities(c1_geometry):
    mesh = generate_mesh(c1_geometry, 0.3, 0)
    m = MeshTriLite(mesh.p, mesh.t)
    A, B, basis, Dxx, Dyy, Dxy, Minv = hfield.assemble_hfield_system(c1_geometry, m)
    A2, B2, _, Dxx2, Dyy2, Dxy2, Minv2 = hfield.assemble_hfield_system_fused(c1_geometry, m)
    scale = abs(A).max()
    assert abs(A - A2).max() <= 1e-15 * scale
    assert abs(B - B2).max() <= 1e-15 * abs(B).max()
    assert abs(Dxy - Dxy2).max() <= 1e-15 * abs(Dxy).max()
    N = basis.N
    assert A.shape == (2 * N, 2 * N) and B.shape == (2 * N, 2 * N)
    assert abs(A - A.T).max() <= 4e-16 * scale                            # A = A^T up to summation order
    assert abs(A[N:, :N] - A[:N, N:].T).max() <= 4e-16 * scale            # A_yx = A_xy^T
    assert abs(B[:N, N:]).max() == 0 and abs(B[:N, :N] - B[N:, N:]).max() == 0
    assert abs(B[:N, :N] - Minv).max() == 0
    # epsilon is two-valued: B entries bounded by the cladding mass
    eps_c = c1_geometry.n_core ** 2
    assert Minv.sum() < np.pi * 32.0 ** 2 and Minv.sum() > np.pi * 32.0 ** 2 / eps_c * 0.9


def test_solve_matches_direct_eigsh_semantics(c1_geometry):
    mesh = generate_mesh(c1_geometry, 0.3, 0)
    m = MeshTriLite(mesh.p, mesh.t)
    tm = {}
    modes, raw = hfield.solve_vectorial_modes(c1_geometry, m, 4, fused=True, return_raw=True, timings=tm)
    assert raw["sigma"] == 26.150716554571733
    assert len(raw["beta_sq"]) == 4 + 12 and tm["n"] == 2 * tm["N_solve"]
    # eigen-residuals and B-orthonormality of what eigsh returned
    A, Bm, V, w = raw["A_int"], raw["B_int"], raw["evecs"], raw["beta_sq"]
    R = A @ V - (Bm @ V) * w
    assert np.abs(R).max() / np.abs(A @ V).max() < 1e-8
    assert np.abs(V.T @ (Bm @ V) - np.eye(len(w))).max() < 1e-10
    # the k returned are the k nearest sigma: no other eigenvalue is closer (count via inertia-free check)
    lu = spla.splu((A - raw["sigma"] * Bm).tocsc())
    assert np.isfinite(lu.solve(np.ones(A.shape[0]))).all()
    # ordering / keys of the mode records (solver_fem.py:222-239)
    ne = [x["n_eff"] for x in modes]
    assert ne == sorted(ne, reverse=True)
    keys = {"n_eff", "beta", "Ex_dofs", "Ey_dofs", "P_x", "P_y", "PDL_dB", "polarization", "confinement",
            "core_overlap", "div_ratio", "is_vectorial", "method"}
    assert set(modes[0]) == keys and modes[0]["method"] == "H-field_V18.10"
    v = np.concatenate([modes[0]["Ex_dofs"], modes[0]["Ey_dofs"]])
    assert abs(np.linalg.norm(v) - 1.0) < 1e-12
