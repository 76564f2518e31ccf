"""CPU checks of the row f3 / f4 oracles (oracle/scalar.py, oracle/cmt.py) and of the host analysis with one unknown
per node.  Closed-form properties: parity of these two rows is otherwise unpinned (the reference holds no test for
them and scikit-fem / config.py cannot run here)."""
import numpy as np
import pytest

from oracle import cmt, scalar
from oracle.p2 import MeshTriLite, P2Basis
from pl_fem_vectoriel_amd import MCFGeometry, _native
from pl_fem_vectoriel_amd.mesh import generate_mesh, unit_square_mesh


class _Uniform:
    """geometry duck-type with a constant permittivity"""
    def __init__(self, eps):
        self.eps, self.k0, self.n_core, self.n_clad = eps, 4.0, 1.5, 1.0
        self.positions, self.core_radii = np.zeros((1, 2)), np.array([0.1])

    def epsilon(self, x, y):
        return np.full(np.shape(x), self.eps, dtype=complex)


def test_scalar_forms_closed_form_properties():
    sq = unit_square_mesh(5)
    om = MeshTriLite(sq.p, sq.t)
    g = _Uniform(2.25)
    for fused in (False, True):
        K, M, Me, basis = scalar.assemble(g, om, fused=fused)
        assert abs(M.sum() - 1.0) < 1e-13                              # sum of the mass matrix = area of the unit square
        assert np.abs(K @ np.ones(basis.N)).max() < 1e-12              # constants are in the kernel of the stiffness
        assert abs(Me - 2.25 * M).max() < 1e-15                        # constant eps: M_eps = eps M
        x, y = basis.doflocs
        u = 0.3 + 1.2 * x - 0.7 * y                                    # P2 reproduces linears: u^T K u = |grad u|^2 area
        assert abs(u @ (K @ u) - (1.2 ** 2 + 0.7 ** 2)) < 1e-12
    assert scalar.shift(g) == -(4.0 * (1.5 - 0.008)) ** 2


def test_scalar_oracle_mode_records():
    g = MCFGeometry(3, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    mesh = generate_mesh(g, 0.35, 0)
    modes, raw = scalar.solve(g, MeshTriLite(mesh.p, mesh.t), n_modes_target=6, return_raw=True)
    assert len(raw["evals"]) == 14 and 0 < len(modes) <= 14            # k = n_modes_target + 8 (solver_fem.py:261)
    assert [m["n_eff"] for m in modes] == sorted((m["n_eff"] for m in modes), reverse=True)
    for m in modes:
        assert set(m) == {"n_eff", "beta", "field_vector", "confinement", "core_overlap", "PDL_dB", "polarization", "is_vectorial"}
        assert g.n_clad < m["n_eff"] < 1.005 * g.n_core and m["beta"] == pytest.approx(g.k0 * m["n_eff"])
        v = m["field_vector"]
        assert v @ (raw["M"] @ v) == pytest.approx(1.0, rel=1e-12)     # M-normalised (solver_fem.py:268)
        assert 0.0 <= m["confinement"] <= 1.0 and m["polarization"] == "scalar" and m["is_vectorial"] is False
    # the three cores are equivalent: the fundamental supermodes sit just below n_core
    assert modes[0]["n_eff"] > 1.45


def test_cmt_oracle_properties():
    sq = unit_square_mesh(4)
    basis = P2Basis(MeshTriLite(sq.p, sq.t))
    rng = np.random.default_rng(3)
    modes = [{"beta": 5.0 + i, "field_vector": rng.standard_normal(basis.N)} for i in range(4)]
    H0 = cmt.rigorous_coupling(modes, modes, _Uniform(2.0), basis, omega=1.0e15)
    assert np.abs(H0 - np.diag([5.0, 6.0, 7.0, 8.0])).max() == 0.0     # eps == mean(eps): no coupling at all
    g = MCFGeometry(1, 0.0, 0.3, 1.535, 1.0, wavelength_um=1.55)
    sq.p[:] = sq.p - 0.5
    basis = P2Basis(MeshTriLite(sq.p, sq.t))
    M, mean = cmt.delta_eps_mass(g, basis)
    qx, qy = basis.qx
    eps = np.real(g.epsilon(qx, qy))
    assert mean == pytest.approx(eps.mean()) and abs(M - M.T).max() < 1e-18
    assert M.sum() == pytest.approx(float(np.sum(basis.dx * (eps - mean))), rel=1e-12)   # 1^T M 1 = integral of (eps - mean)
    H = cmt.rigorous_coupling(modes, modes, g, basis, omega=2.0)
    assert np.abs(H - H.T).max() == 0.0 and H.dtype == complex
    H2 = cmt.rigorous_coupling(modes, modes, g, basis, omega=4.0)
    off = ~np.eye(4, dtype=bool)
    assert np.allclose(H2[off], 2.0 * H[off], rtol=1e-14)               # linear in omega, diagonal = beta


def test_host_analysis_with_one_unknown_per_node():
    g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    mesh = generate_mesh(g, 0.5, 0)
    S1 = _native.Symbolic(mesh.p, mesh.t, dofs_per_node=1, dirichlet=False)
    S2 = _native.Symbolic(mesh.p, mesh.t)
    assert (S1.dofs_per_node, S2.dofs_per_node) == (1, 2)
    assert S1.N == S2.N and S1.nsolve == S1.N and S2.nsolve < S2.N      # natural boundary keeps every node
    assert S1.array("bmask").sum() == 0 and np.array_equal(S1.array("interior"), np.arange(S1.N))
    for S, q in ((S1, 16), (S2, 8)):                                    # fronts padded to 16 DOFs
        fs, fb = S.array("fs"), S.array("fb")
        assert (fs % q == 0).all() and (fb % q == 0).all()
        foff = S.array("foff")
        m = S.dofs_per_node * (fs + fb)
        s2 = (S.dofs_per_node * fs).astype(np.int64)
        assert np.array_equal(np.diff(foff), s2 * (2 * m.astype(np.int64) - s2))     # [F11; F21] m x s2 + Z^T s2 x b2
        # front-order maps: every kept node has exactly one owned slot; prow inverts the child maps
        fp, fn, fst = S.array("fnode_ptr"), S.array("fnodes"), S.array("fs_true")
        npos, prow, c0, c1 = S.array("npos"), S.array("prow"), S.array("cinv0"), S.array("cinv1")
        assert (npos >= 0).sum() == S.nsolve and len(np.unique(npos[npos >= 0])) == S.nsolve
        nf = len(fs)
        for f in (0, 1, 2, 5, nf // 2 - 1):
            for q_ in range(fst[f]):
                assert npos[fn[fp[f] + q_]] == 2 * fp[f] + S.dofs_per_node * q_
            for ch, cinv in ((2 * f + 1, c0), (2 * f + 2, c1)):
                for qp in range(fs[f] + fb[f]):
                    c = cinv[fp[f] + qp]
                    if c >= 0:
                        assert prow[fp[ch] + fs[ch] + c] == qp
                        assert fn[fp[ch] + fs[ch] + c] == fn[fp[f] + qp]   # the same mesh node on both sides
    with pytest.raises(ValueError):
        _native.Symbolic(mesh.p, mesh.t, dofs_per_node=3)
