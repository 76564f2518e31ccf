"""Known-answer test for the eps-weighted forms on ONE element that a core interface cuts (the only
mesh-dependent choice of the nine bilinear forms is which quadrature points of such an element see eps_core --
``1 / np.real(eps_fn(x, y))`` at the points of the degree-4 rule, reference ``solver_fem.py:132-150``,
``geometry_unified.py:325-336``: closed discs).

Element: v0 = (0, 0), v1 = (h, 0), v2 = (h, h); one core, centre (0, 0), radius 0.65 h; eps_core = 4, eps_clad = 1.
Map x = h (xi + eta), y = h eta, |det J| = h^2.  Distance^2 / h^2 of the six points (a = 0.445948490915965,
b = 0.091576213509771; point order of oracle/p2.py):
    (a, a)       5 a^2               = 0.99435   outside        (b, b)       5 b^2               = 0.04193   INSIDE
    (1-2a, a)    (1-a)^2 + a^2       = 0.50584   outside        (1-2b, b)    (1-b)^2 + b^2       = 0.83362   outside
    (a, 1-2a)    (1-a)^2 + (1-2a)^2  = 0.31866   INSIDE         (b, 1-2b)    (1-b)^2 + (1-2b)^2  = 1.49247   outside
against 0.65^2 = 0.4225: exactly two points, one of each weight class, lie in the core.  Everything below follows from
that by hand: e.g. sum_ij M_epsinv[i][j] = h^2 (wA (1 + 1 + 1/4) + wB (1/4 + 1 + 1)) by the partition of unity.
The expected element matrices are built here from the formulas of SURVEY.md row a2 (vi) written out again (not imported
from the oracle), checked against the hand facts, and then both the oracle and the HIP kernel must reproduce them."""
import numpy as np
import pytest

from oracle import hfield
from oracle.p2 import MeshTriLite, P2Basis

A_, B_ = 0.445948490915965, 0.091576213509771
WA, WB = 0.223381589678011 / 2.0, 0.109951743655322 / 2.0
PTS = [(A_, A_), (1 - 2 * A_, A_), (A_, 1 - 2 * A_), (B_, B_), (1 - 2 * B_, B_), (B_, 1 - 2 * B_)]
WTS = [WA, WA, WA, WB, WB, WB]
H, RHO, EPS_CORE, EPS_CLAD = 0.25, 0.65, 4.0, 1.0
INSIDE = [False, False, True, True, False, False]          # the table in the docstring


def _phi(x, y):
    return [1 - 3 * x - 3 * y + 2 * x * x + 4 * x * y + 2 * y * y, 2 * x * x - x, 2 * y * y - y, 4 * x - 4 * x * x - 4 * x * y, 4 * x * y,
            4 * y - 4 * x * y - 4 * y * y]


def _dphi(x, y):                                           # (d/dxi, d/deta)
    return [(-3 + 4 * x + 4 * y, -3 + 4 * x + 4 * y), (4 * x - 1, 0.0), (0.0, 4 * y - 1), (4 - 8 * x - 4 * y, -4 * x), (4 * y, 4 * x),
            (-4 * y, 4 - 4 * x - 8 * y)]


def expected_forms():
    """out[name][i][j], i = test function (row), j = trial function (column), as asm() lays them out."""
    names = ("kxx", "kyy", "kxy", "kyx", "div_xx", "div_yy", "div_xy", "mass", "mass_eps_inv")
    out = {k: np.zeros((6, 6)) for k in names}
    for (xi, eta), w, inside in zip(PTS, WTS, INSIDE):
        x, y = H * (xi + eta), H * eta
        assert (x * x + y * y <= (RHO * H) ** 2) == inside                      # the classification done by hand above
        ie = 1.0 / (EPS_CORE if inside else EPS_CLAD)
        dx_ = w * H * H
        ph = _phi(xi, eta)
        # J = [[h, h], [0, h]]  =>  J^-T = (1/h) [[1, 0], [-1, 1]]:  d/dx = (1/h) d/dxi,  d/dy = (1/h) (d/deta - d/dxi)
        g = [(d[0] / H, (d[1] - d[0]) / H) for d in _dphi(xi, eta)]
        for i in range(6):          # v = test
            for j in range(6):      # u = trial
                ux, uy, vx, vy = g[j][0], g[j][1], g[i][0], g[i][1]
                out["kxx"][i, j] += ie * uy * vy * dx_
                out["kyy"][i, j] += ie * ux * vx * dx_
                out["kxy"][i, j] += -ie * uy * vx * dx_
                out["kyx"][i, j] += -ie * ux * vy * dx_
                out["div_xx"][i, j] += ux * vx * dx_
                out["div_yy"][i, j] += uy * vy * dx_
                out["div_xy"][i, j] += ux * vy * dx_
                out["mass"][i, j] += ph[j] * ph[i] * dx_
                out["mass_eps_inv"][i, j] += ie * ph[j] * ph[i] * dx_
    return out


class OneDisc:
    """Geometry duck type: one closed disc at the origin (geometry_unified.py:325-336 without the PML factor)."""
    k0 = 0.0
    n_core, n_clad = 2.0, 1.0
    core_positions = np.array([[0.0, 0.0]])
    core_radii = np.array([RHO * H])

    def epsilon(self, x, y):
        return np.where(x * x + y * y <= (RHO * H) ** 2, EPS_CORE, EPS_CLAD) + 0j


def test_hand_facts_of_the_expected_matrices():
    E = expected_forms()
    total = H * H * (WA * (1 + 1 + 0.25) + WB * (0.25 + 1 + 1))
    assert abs(E["mass_eps_inv"].sum() - total) < 1e-16
    assert abs(E["mass"].sum() - 0.5 * H * H) < 1e-16
    # phi1 = 2 xi^2 - xi has d/dx = (4 xi - 1) / h, d/dy = -(4 xi - 1) / h on this element:
    s = sum(w * (1 / (EPS_CORE if ins else EPS_CLAD)) * (4 * xi - 1) ** 2 for (xi, _e), w, ins in zip(PTS, WTS, INSIDE))
    assert abs(E["kxx"][1, 1] - s) < 1e-15 and abs(E["kyy"][1, 1] - s) < 1e-15 and abs(E["kxy"][1, 1] - s) < 1e-15
    for k in ("kxx", "kyy", "kxy", "kyx", "div_xx", "div_yy", "div_xy"):
        assert np.abs(E[k].sum(axis=0)).max() < 1e-14 and np.abs(E[k].sum(axis=1)).max() < 1e-14
    np.testing.assert_allclose(E["kyx"], E["kxy"].T, atol=1e-16)
    # the interface is felt: with all six points in the cladding the eps-weighted mass would be the plain mass
    assert abs(E["mass_eps_inv"].sum() - E["mass"].sum()) > 0.05 * E["mass"].sum()


def test_oracle_forms_on_the_interface_cut_element():
    mesh = MeshTriLite(np.array([[0.0, H, H], [0.0, 0.0, H]]), np.array([[0], [1], [2]]))
    em = hfield.element_matrices(OneDisc(), P2Basis(mesh))
    E = expected_forms()
    for k, M in E.items():
        # element_matrices stores [e, i, j] with i = test, j = trial (hfield.py: bil(a_trial, b_test, w))
        np.testing.assert_allclose(em[k][0], M, rtol=0, atol=2e-15 * max(1.0, np.abs(M).max()), err_msg=k)


@pytest.mark.gpu
def test_hip_element_matrices_on_the_interface_cut_element(gpu_device, built_library):
    from pl_fem_vectoriel_amd import _native
    from pl_fem_vectoriel_amd.mesh import unit_square_mesh
    mesh = unit_square_mesh(4)                      # element 0 = vertices 0, 1, 6 = (0, 0), (h, 0), (h, h) with h = 1/4
    assert mesh.t[:, 0].tolist() == [0, 1, 6] and np.allclose(mesh.p[:, [0, 1, 6]], [[0, H, H], [0, 0, H]])
    sym = _native.Symbolic(mesh.p, mesh.t)
    ctx = _native.Context(sym, gpu_device, max_ncv=45)
    E = expected_forms()
    for k0 in (0.0, 1.7):
        ctx.assemble(np.array([[0.0, 0.0, RHO * H]]), EPS_CORE, EPS_CLAD, k0, 1.0)
        el = ctx.debug_copy("elem", 0, 8 * 36).reshape(8, 6, 6)       # Axx Axy Ayx Ayy Minv Dxx Dxy Dyy, [i][j]
        want = {0: E["kxx"] + E["div_xx"] - k0 ** 2 * E["mass"], 1: E["kxy"] + E["div_xy"], 2: E["kyx"] + E["div_xy"].T,
                3: E["kyy"] + E["div_yy"] - k0 ** 2 * E["mass"], 4: E["mass_eps_inv"], 5: E["div_xx"], 6: E["div_xy"], 7: E["div_yy"]}
        for b, M in want.items():
            np.testing.assert_allclose(el[b], M, rtol=0, atol=1e-14 * max(1.0, np.abs(M).max()), err_msg=f"block {b}")
    ctx.close()
