"""Known-answer tests pinning the scikit-fem restatement in oracle/p2.py (the
reference ships no tests for this path; these closed-form facts stand in)."""
import numpy as np
import pytest

from oracle import hfield
from oracle.p2 import DPHI_Q, PHI_Q, QUAD_W, QUAD_X, MeshTriLite, P2Basis, p2_basis
from pl_fem_vectoriel_amd.mesh import TriMesh, generate_mesh, unit_square_mesh


class UnitEps:
    k0 = 1.0

    def epsilon(self, x, y):
        return np.ones_like(x) + 0j


def test_basis_is_nodal_and_partition_of_unity():
    nodes = np.array([[0, 0], [1, 0], [0, 1], [.5, 0], [.5, .5], [0, .5]]).T
    phi, dphi = p2_basis(nodes[0], nodes[1])
    np.testing.assert_allclose(phi, np.eye(6), atol=1e-15)
    x, y = np.random.default_rng(0).uniform(0, .5, (2, 20))
    phi, dphi = p2_basis(x, y)
    np.testing.assert_allclose(phi.sum(0), 1.0, atol=1e-14)
    np.testing.assert_allclose(dphi.sum(0), 0.0, atol=1e-13)


def test_quadrature_is_the_degree4_rule():
    assert abs(QUAD_W.sum() - 0.5) < 1e-15
    from math import factorial
    for a in range(6):
        for b in range(6 - a):
            exact = factorial(a) * factorial(b) / factorial(a + b + 2)
            got = np.sum(QUAD_W * QUAD_X[0] ** a * QUAD_X[1] ** b)
            if a + b <= 4:
                assert abs(got - exact) < 1e-15, (a, b)
    # NOT exact at degree 5 (discriminates this rule from a higher-order one)
    assert abs(np.sum(QUAD_W * QUAD_X[0] ** 5) - 1 / 42) > 1e-5


def test_reference_mass_matrix_closed_form():
    m = MeshTriLite(np.array([[0, 1, 0], [0, 0, 1.0]]), np.array([[0], [1], [2]]))
    em = hfield.element_matrices(UnitEps(), P2Basis(m))
    M = np.array([[6, -1, -1, 0, -4, 0], [-1, 6, -1, 0, 0, -4], [-1, -1, 6, -4, 0, 0],
                  [0, 0, -4, 32, 16, 16], [-4, 0, 0, 16, 32, 16], [0, -4, 0, 16, 16, 32]]) * (0.5 / 180)
    np.testing.assert_allclose(em["mass"][0], M, atol=1e-16)
    np.testing.assert_allclose(em["mass_eps_inv"][0], M, atol=1e-16)
    for k in ("kxx", "kyy", "kxy", "kyx", "div_xx", "div_yy", "div_xy"):
        assert np.abs(em[k][0].sum(axis=1)).max() < 5e-15      # gradients of constants vanish
        assert np.abs(em[k][0].sum(axis=0)).max() < 5e-15
    np.testing.assert_allclose(em["kyx"][0], em["kxy"][0].T, atol=1e-16)


def test_numbering_matches_scikit_fem_convention():
    # two triangles sharing an edge, vertices given unsorted
    p = np.array([[0, 1, 1, 0], [0, 0, 1, 1.0]])
    t = np.array([[2, 0], [0, 2], [1, 3]])
    m = MeshTriLite(p, t)
    np.testing.assert_array_equal(m.t, [[0, 0], [1, 2], [2, 3]])           # sort_t
    # edges in lexicographic order of (min, max): (0,1) (0,2) (0,3) (1,2) (2,3)
    np.testing.assert_array_equal(m.facets, [[0, 0, 0, 1, 2], [1, 2, 3, 2, 3]])
    b = P2Basis(m)
    assert b.N == 4 + 5
    # element_dofs rows 3..5 = edges (0,1), (1,2), (0,2) of the sorted triangle
    np.testing.assert_array_equal(b.element_dofs[:, 0], [0, 1, 2, 4 + 0, 4 + 3, 4 + 1])
    np.testing.assert_array_equal(b.element_dofs[:, 1], [0, 2, 3, 4 + 1, 4 + 4, 4 + 2])
    # boundary: all vertices + the four outer edges; the shared diagonal (0,2) is interior
    np.testing.assert_array_equal(b.get_dofs().all(), [0, 1, 2, 3, 4, 6, 7, 8])
    np.testing.assert_allclose(b.doflocs[:, 4 + 1], [0.5, 0.5])


@pytest.mark.parametrize("n", [2, 5])
def test_patch_and_area(n):
    mesh = unit_square_mesh(n)
    m = MeshTriLite(mesh.p, mesh.t)
    b = P2Basis(m)
    A, B, basis, Dxx, Dyy, Dxy, Minv = hfield.assemble_hfield_system_fused(UnitEps(), m)
    N = b.N
    assert abs(Minv.sum() - 1.0) < 1e-13                                   # sum M = area
    one = np.ones(N)
    assert np.abs(Dxx @ one).max() < 1e-12 and np.abs(Dyy @ one).max() < 1e-12
    x, y = b.doflocs
    q = 1 + 2 * x - y + 0.5 * x * x + x * y - 2 * y * y                   # quadratic: reproduced exactly by P2
    gx = 2 + x + y
    # energy of dq/dx: int (dq/dx)^2 = q^T Dxx q
    exact = 4 + 1 / 3 + 1 / 3 + 2 + 2 + 0.5
    assert abs(q @ (Dxx @ q) - exact) < 1e-12
    assert abs((A - A.T)).max() < 1e-12
    assert (B.diagonal() > 0).all()


def test_refinement_conserves_area_and_matches_product_mesh():
    mesh = unit_square_mesh(3)
    a = MeshTriLite(mesh.p, mesh.t).refined(2)
    bmesh = mesh.refined(2)
    np.testing.assert_array_equal(a.t, bmesh.t)
    np.testing.assert_array_equal(a.p, bmesh.p)
    assert a.t.shape[1] == 18 * 16
    assert abs(P2Basis(a).absdet.sum() / 2 - 1.0) < 1e-14


def test_synthetic_mesh_sizes_match_survey(c1_geometry):
    m0 = generate_mesh(c1_geometry, 1.0, 0)
    assert (m0.nvertices, m0.nelements) == (5691, 11313)                   # 25 zero-area triangles dropped
    b = P2Basis(MeshTriLite(m0.p, m0.t))
    assert b.N == 22694 and len(b.get_dofs().all()) == 134
    assert b.absdet.min() > 1e-10
    with pytest.raises(ValueError):
        TriMesh(m0.p, np.array([[0], [1], [99999]]))
