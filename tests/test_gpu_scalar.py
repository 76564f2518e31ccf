"""Row f3: ScalarHelmholtzSolver.solve (reference solver_fem.py:245-276) on the GPU, one
unknown per P2 node through the same front tree / sweeps / Lanczos driver, against the oracle (oracle/scalar.py +
SciPy eigsh with the reference's arguments).  Tolerances as for the vectorial path: |dn_eff| < 5e-5, field L2 < 1e-6."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import scalar
from oracle.compare import column_errors
from oracle.p2 import MeshTriLite
from pl_fem_vectoriel_amd import MCFGeometry, _native
from pl_fem_vectoriel_amd.geometry import ARRANGEMENTS
from pl_fem_vectoriel_amd.mesh import generate_mesh, unit_square_mesh
from pl_fem_vectoriel_amd.solver_fem import ScalarHelmholtzSolver, _core_table

pytestmark = pytest.mark.gpu
N_EFF_TOL, FIELD_TOL = 5e-5, 1e-6


def test_scalar_pencil_blocks_spmv_and_solve(c1_geometry, gpu_device, built_library):
    import torch
    g = c1_geometry
    mesh = generate_mesh(g, 0.5, 0)
    om = MeshTriLite(mesh.p, mesh.t)
    K, M, Me, basis = scalar.assemble(g, om, eliminate_zeros=False)
    A = (K - g.k0 ** 2 * Me).tocsr()
    sym = _native.Symbolic(mesh.p, mesh.t, dofs_per_node=1, dirichlet=False, leaf_elems=24)
    assert sym.dofs_per_node == 1 and sym.nsolve == sym.N == basis.N
    assert (sym.array("fs") % 16 == 0).all() and sym.array("bmask").sum() == 0
    ctx = _native.Context(sym, gpu_device, max_ncv=65)
    assert ctx.n2 == sym.N
    with pytest.raises(ValueError):
        ctx.assemble(_core_table(g), g.n_core ** 2, g.n_clad ** 2, g.k0, 1.0)          # vectorial assembly on a scalar context
    ctx.assemble_scalar(_core_table(g), g.n_core ** 2, g.n_clad ** 2, g.k0)
    rowptr, colind = sym.array("rowptr"), sym.array("colind")
    N = sym.N
    Ag = sp.csr_matrix((ctx.block_values("Axx"), colind, rowptr), shape=(N, N))
    Mg = sp.csr_matrix((ctx.block_values("Minv"), colind, rowptr), shape=(N, N))
    assert abs(Ag - A).max() <= 1e-12 * abs(A).max()
    assert abs(Mg - M).max() <= 1e-13 * abs(M).max()
    x = np.random.default_rng(0).standard_normal(N)
    xd = torch.from_numpy(x).cuda()
    for which, R in (("A", A), ("B", M)):
        y = ctx.spmv(which, xd).cpu().numpy()
        assert np.abs(y - R @ x).max() <= 1e-13 * (abs(R) @ np.abs(x)).max()
    sigma = scalar.shift(g)
    ctx.factor(sigma)
    assert ctx.timings()["pivot_perturbations"] == 0
    lu = spla.splu((A - sigma * M).tocsc())
    xs = lu.solve(x)
    x0 = ctx.solve(xd, 0).cpu().numpy()
    assert np.linalg.norm(x0 - xs) / np.linalg.norm(xs) < 1e-9
    ctx.close()


@pytest.mark.parametrize("arrangement", ["hexagonal_1plus6_7", "triangular_3", "square_2x2_4"])
def test_scalar_modes_match_oracle(arrangement, gpu_device, built_library):
    n, variant = ARRANGEMENTS[arrangement]
    g = MCFGeometry(n, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55, variant=variant)
    mesh = generate_mesh(g, 0.5, 0)
    solver = ScalarHelmholtzSolver(g, device=gpu_device)
    modes = solver.solve(mesh, n_modes_target=10)
    ref, raw = scalar.solve(g, MeshTriLite(mesh.p, mesh.t), 10, return_raw=True)
    st = solver.last_stats
    assert st["n_req"] == 18 and st["nconv"] == 18 and st["true_residual"] < ScalarHelmholtzSolver.RESIDUAL_TOL and st["refined"] is False and st["pivot_perturbations"] == 0
    assert len(modes) == len(ref) > 0
    assert [m["n_eff"] for m in modes] == sorted((m["n_eff"] for m in modes), reverse=True)
    for a, b in zip(modes, ref):
        assert set(a) == set(b)
        assert abs(a["n_eff"] - b["n_eff"]) < N_EFF_TOL and abs(a["beta"] - b["beta"]) < 1e-4
        assert a["polarization"] == "scalar" and a["is_vectorial"] is False and a["PDL_dB"] == 0.0
        assert a["field_vector"].shape == b["field_vector"].shape
    # fields: sign-invariant per mode, subspace distance inside (near-)degenerate clusters; both M-normalised
    V = np.array([m["field_vector"] for m in modes])
    U = np.array([m["field_vector"] for m in ref]).T
    Mm = raw["M"]
    assert np.abs(np.einsum("ij,ji->i", V, Mm @ V.T) - 1.0).max() < 1e-10           # v.M v = 1 (solver_fem.py:268)
    lam = -np.array([(m["beta"]) ** 2 for m in ref])
    nrm = np.linalg.norm(U, axis=0)
    err = column_errors((V / np.linalg.norm(V, axis=1)[:, None]).T, U / nrm, lam, 1e-5)
    assert err.max() < FIELD_TOL
    gaps = np.abs(np.diff(lam)) / np.abs(lam[1:])
    iso = np.ones(len(lam), bool)
    iso[:-1] &= gaps > 1e-5
    iso[1:] &= gaps > 1e-5
    for a, b, single in zip(modes, ref, iso):
        if single:
            assert abs(a["confinement"] - b["confinement"]) < 1e-6 and a["core_overlap"] == a["confinement"]
    # the records' consumer (row f2, scalar route: losses.py:828-865): same loss columns from the GPU records and from the
    # oracle's.  (The crosstalk estimate of scalar records overlaps the raw field_vector arrays, whose sign and basis
    # inside a degenerate pair are the eigensolver's choice: finite, not compared.)
    from pl_fem_vectoriel_amd.losses import LossCalculator
    got = LossCalculator.calculate_physical_losses(modes, g, "mux", 1550.0)
    want = LossCalculator.calculate_physical_losses(ref, g, "mux", 1550.0)
    assert got["success"] and got["is_vectorial"] is False and got["n_modes_used"] == len(modes)
    if iso.all():
        for key in ("IL_dB", "MDL_dB", "PDL_dB", "radiation_loss_dB_per_m", "avg_confinement"):
            assert abs(got[key] - want[key]) <= 1e-6 * max(1.0, abs(want[key])), key
    assert np.isfinite(got["crosstalk_dB"])


def test_scalar_tiny_mesh_clamps_the_request(gpu_device, built_library):
    """k = min(n_modes_target + 8, N - 4) (solver_fem.py:261) on a 2 x 2 square: N = 25 -> 21 pairs requested."""
    g1 = MCFGeometry(1, 0.0, 0.3, 1.535, 1.0, wavelength_um=1.55)
    sq = unit_square_mesh(2)
    sq.p[:] = sq.p - 0.5
    solver = ScalarHelmholtzSolver(g1, device=gpu_device)
    modes = solver.solve(sq, n_modes_target=40)
    assert solver.last_stats["n_req"] == 21 == solver.last_stats["N"] - 4 and solver.last_stats["nconv"] == 21
    ref = scalar.solve(g1, MeshTriLite(sq.p, sq.t), 40)
    assert len(modes) == len(ref)
    for a, b in zip(modes, ref):
        assert abs(a["n_eff"] - b["n_eff"]) < N_EFF_TOL


def test_cmt_rigorous_coupling_matches_oracle(c1_geometry, gpu_device, built_library):
    """Row f4: CoupledModeTheory._compute_rigorous_coupling (reference config.py:274-322) from the scalar modes of a GPU
    solve: H against the oracle's restatement on the same fields, plus the closed-form properties."""
    from oracle import cmt as ocmt
    from oracle.p2 import P2Basis
    from pl_fem_vectoriel_amd.cmt import CoupledModeTheory
    g = c1_geometry
    mesh = generate_mesh(g, 0.5, 0)
    solver = ScalarHelmholtzSolver(g, device=gpu_device)
    modes = solver.solve(mesh, n_modes_target=6)
    solver.clear_cache()
    omega = 2 * np.pi * 2.99792458e14 / 1.55
    theory = CoupledModeTheory(omega, "rigorous", device=gpu_device)
    H = theory._compute_rigorous_coupling(modes, modes, g, mesh)
    basis = P2Basis(MeshTriLite(mesh.p, mesh.t))
    Href = ocmt.rigorous_coupling(modes, modes, g, basis, omega)
    _, mean = ocmt.delta_eps_mass(g, basis)
    assert abs(theory.last_stats["eps_mean"] - mean) < 1e-13
    assert H.shape == Href.shape == (len(modes), len(modes)) and H.dtype == complex
    assert np.array_equal(np.diag(H), [m["beta"] for m in modes])
    assert np.abs(H - H.T).max() == 0.0
    scale = np.abs(Href - np.diag(np.diag(Href))).max()
    assert np.abs(H - Href).max() <= 1e-10 * scale
    with pytest.raises(ValueError):
        CoupledModeTheory(omega, "exact")
    with pytest.raises(ValueError):
        theory._compute_rigorous_coupling(modes, modes[:-1], g, mesh)


def test_scalar_solver_at_north_star_size(c1_geometry, gpu_device, built_library):
    """The scalar solver on the C1 mesh (N = 90 639 unknowns, 12 tree levels, long fronts -> every sweep kernel form
    with one unknown per node): n_eff against the oracle, every pair an eigenpair of the assembled pencil."""
    g = c1_geometry
    mesh = generate_mesh(g, 1.0, 1)
    solver = ScalarHelmholtzSolver(g, device=gpu_device)
    modes = solver.solve(mesh, n_modes_target=10)
    st = solver.last_stats
    assert st["N"] == 90639 and st["nconv"] == 18 and st["refined"] is False and st["n_block_solves"] > 0
    assert st["true_residual"] < ScalarHelmholtzSolver.RESIDUAL_TOL and st["pivot_perturbations"] == 0
    ref = scalar.solve(g, MeshTriLite(mesh.p, mesh.t), 10)
    assert len(modes) == len(ref) > 0
    assert max(abs(a["n_eff"] - b["n_eff"]) for a, b in zip(modes, ref)) < N_EFF_TOL
    solver.clear_cache()
