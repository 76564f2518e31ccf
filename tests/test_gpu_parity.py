"""GPU parity tests: the HIP path, called through the C-ABI (ctypes), against the oracle on identical
meshes.  Tolerances: north_star asks |dn_eff| < 5e-5 and field L2 < 1e-6; the matrices themselves are
compared to 1e-13 of the assembled magnitude."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import front_emulation as fe
from oracle import hfield
from oracle.compare import column_errors, mode_field_errors
from oracle.p2 import MeshTriLite, P2Basis
from pl_fem_vectoriel_amd import MCFGeometry, _native
from pl_fem_vectoriel_amd.mesh import generate_mesh, unit_square_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver, _core_table, shift_estimate

pytestmark = pytest.mark.gpu

N_EFF_TOL = 5e-5      # north_star
FIELD_TOL = 1e-6      # north_star


class Problem:
    def __init__(self, g, mesh, device, leaf_elems=0):
        import torch
        self.torch = torch
        self.g, self.mesh = g, mesh
        self.sym = _native.Symbolic(mesh.p, mesh.t, leaf_elems=leaf_elems)
        self.ctx = _native.Context(self.sym, device, max_ncv=65)
        self.ctx.assemble(_core_table(g), g.n_core ** 2, g.n_clad ** 2, g.k0, 1.0)
        self.om = MeshTriLite(mesh.p, mesh.t)
        self.basis = P2Basis(self.om)
        self.em = hfield.element_matrices(g, self.basis)
        self.A, self.B, _, self.Dxx, self.Dyy, self.Dxy, self.Minv = \
            hfield.assemble_hfield_system_fused(g, self.om, eliminate_zeros=False)
        self.N = self.basis.N
        self.A_int, self.B_int, self.interior = hfield.restrict_interior(self.A, self.B, self.basis)
        self.idx = np.concatenate([self.interior, self.interior + self.N])
        self.sigma = shift_estimate(g)

    def embed(self, v):
        full = np.zeros(2 * self.N)
        full[self.idx] = v
        return self.torch.from_numpy(full).cuda()


@pytest.fixture(scope="module")
def small(c1_geometry, gpu_device, built_library):
    return Problem(c1_geometry, generate_mesh(c1_geometry, 0.5, 0), gpu_device, leaf_elems=24)


@pytest.fixture(scope="module")
def medium(c1_geometry, gpu_device, built_library):
    """C3 ladder rung L=0 of the north-star geometry (N = 22 694), contains near-degenerate triangles."""
    return Problem(c1_geometry, generate_mesh(c1_geometry, 1.0, 0), gpu_device)


def _abs_assembled(P, name_terms):
    """Sum of |element entries| per CSR slot: the magnitude rounding errors are measured against."""
    ed = P.basis.element_dofs
    rows = np.broadcast_to(ed.T[:, :, None], (ed.shape[1], 6, 6)).ravel()
    cols = np.broadcast_to(ed.T[:, None, :], (ed.shape[1], 6, 6)).ravel()
    # per-element scale = largest entry of that element's matrices: inside a sliver element an entry
    # can be a cancelling sum of quadrature terms as large as the element's largest entry
    mag = sum(np.abs(t) for t in name_terms)
    mag = np.broadcast_to(mag.max(axis=(1, 2), keepdims=True), mag.shape)
    return sp.coo_matrix((mag.ravel(), (rows, cols)), shape=(P.N, P.N)).tocsr()


@pytest.mark.parametrize("prob", ["small", "medium"])
def test_assembled_blocks_match_oracle(prob, request):
    P = request.getfixturevalue(prob)
    N, em, k0sq = P.N, P.em, P.g.k0 ** 2
    rowptr, colind = P.sym.array("rowptr"), P.sym.array("colind")
    refs = {
        "Axx": (P.A[:N, :N], [em["kxx"], em["div_xx"], k0sq * em["mass"]]),
        "Axy": (P.A[:N, N:], [em["kxy"], em["div_xy"]]),
        "Ayx": (P.A[N:, :N], [em["kyx"], em["div_xy"]]),
        "Ayy": (P.A[N:, N:], [em["kyy"], em["div_yy"], k0sq * em["mass"]]),
        "Minv": (P.Minv, [em["mass_eps_inv"]]), "Dxx": (P.Dxx, [em["div_xx"]]),
        "Dxy": (P.Dxy, [em["div_xy"]]), "Dyy": (P.Dyy, [em["div_yy"]]),
    }
    for name, (R, terms) in refs.items():
        G = sp.csr_matrix((P.ctx.block_values(name), colind, rowptr), shape=(N, N))
        mag = _abs_assembled(P, terms)
        D = abs(G - R)
        D.eliminate_zeros()
        # |diff| <= 1e-13 * (sum over contributing elements of the element's largest entry), slot by slot
        viol = D - 1e-13 * mag
        assert viol.max() <= 0.0, (name, D.max(), mag.max())


def test_reference_surface_assemble_hfield_system(small, gpu_device):
    solver = TrueVectorialMaxwellSolver(small.g, device=gpu_device)
    A, B, basis, Dxx, Dyy, Dxy, Minv = solver.assemble_hfield_system(small.mesh)
    N = small.N
    assert basis.N == N and A.shape == (2 * N, 2 * N) and B.shape == (2 * N, 2 * N)
    np.testing.assert_array_equal(basis.doflocs, small.basis.doflocs)
    np.testing.assert_array_equal(basis.get_dofs().all(), small.basis.get_dofs().all())
    np.testing.assert_array_equal(basis.element_dofs, small.basis.element_dofs)
    sc = abs(small.A).max()
    assert abs(A - small.A).max() < 1e-12 * sc
    assert abs(B - small.B).max() < 1e-13 * abs(small.B).max()
    assert abs(Dxy - small.Dxy).max() < 1e-12 * abs(small.Dxy).max()
    assert abs(Minv - small.Minv).max() < 1e-13 * abs(small.Minv).max()
    assert abs(A - A.T).max() < 1e-12 * sc


def test_spmv_matches_restricted_pencil(medium):
    P = medium
    x = np.random.default_rng(0).standard_normal(len(P.idx))
    xd = P.embed(x)
    for which, M in (("A", P.A_int), ("B", P.B_int)):
        y = P.ctx.spmv(which, xd).cpu().numpy()
        ref = M @ x
        assert np.abs(y[P.idx] - ref).max() <= 1e-13 * (abs(M) @ np.abs(x)).max()
        assert np.abs(np.delete(y, P.idx)).max() == 0.0            # Dirichlet rows masked


def test_fronts_match_numpy_emulation(small):
    P = small
    T = fe.FrontTree(P.sym)
    Ke = fe.element_K(P.em, P.g.k0 ** 2, P.sigma)
    Fs, Ds = fe.factor(T, Ke)
    P.ctx.factor(P.sigma)
    P.ctx.synchronize()
    assert P.ctx.timings()["pivot_perturbations"] == 0
    for f in [0, 1, 2, 5, 11, T.leaf0 - 1, T.leaf0, T.leaf0 + 7, T.nf - 1] + list(range(17, T.nf, 97)):
        m, s2 = T.m(f), T.s2(f)
        # the Schur complement of a front lives in the arena of its tree level until the level two above it reuses the
        # arena: after a complete factorisation only the two fronts of level 1 still have theirs
        keep_s = f in (1, 2)
        Fg = T.device_front(P.ctx, f, with_schur=keep_s)
        blocks = [("F11", Fg[:s2, :s2], Fs[f][:s2, :s2]), ("Z", Fg[s2:, :s2], Fs[f][s2:, :s2]), ("ZT", Fg[:s2, s2:], Fs[f][:s2, s2:])]
        if keep_s:       # maintained in its lower triangle only (symmetric)
            blocks.append(("S", np.tril(Fg[s2:, s2:]), np.tril(Fs[f][s2:, s2:])))
        for name, a, b in blocks:
            if a.size:
                assert np.abs(a - b).max() <= 1e-8 * max(np.abs(b).max(), 1e-300), (f, name)
        if s2:
            dg = P.ctx.debug_copy("delta", 2 * 2 * T.fptr[f], 2 * s2).reshape(s2, 2)     # D^-1: (diagonal, off-diagonal) per row
            assert np.abs(dg - Ds[f]).max() <= 1e-8 * np.abs(Ds[f]).max(), f


@pytest.mark.parametrize("prob", ["small", "medium"])
def test_shift_invert_solve_matches_splu(prob, request):
    P = request.getfixturevalue(prob)
    P.ctx.factor(P.sigma)
    K = (P.A_int - P.sigma * P.B_int).tocsc()
    lu = spla.splu(K)
    b = np.random.default_rng(1).standard_normal(len(P.idx))
    xs = lu.solve(b)
    x0 = P.ctx.solve(P.embed(b), 0).cpu().numpy()
    x1 = P.ctx.solve(P.embed(b), 1).cpu().numpy()
    assert np.abs(np.delete(x0, P.idx)).max() == 0.0
    assert np.linalg.norm(x0[P.idx] - xs) / np.linalg.norm(xs) < 1e-9
    assert np.linalg.norm(x1[P.idx] - xs) / np.linalg.norm(xs) < 1e-9
    r0 = np.linalg.norm(K @ x0[P.idx] - b) / np.linalg.norm(b)
    r1 = np.linalg.norm(K @ x1[P.idx] - b) / np.linalg.norm(b)
    assert r1 <= max(r0, 1e-11) and r1 < 1e-8


def _match_fields(V, U, w, gap_tol=1e-4):
    """Sign-invariant per-mode L2 error; subspace distance inside clusters of eigenvalues (oracle/compare.py)."""
    return column_errors(V, U, np.asarray(w), gap_tol)


def test_eigenpairs_match_scipy_eigsh(medium):
    """k = 22 pairs nearest sigma vs eigsh with the reference's arguments (solver_fem.py:197)."""
    P = medium
    k, ncv = 22, 45
    P.ctx.factor(P.sigma)
    evals, evecs, st = P.ctx.lanczos(k, ncv, 1e-10, 12000, P.sigma)
    assert st["nconv"] == k
    assert st["restarts"] >= 1          # a 48-column basis at this tolerance: the thick-restart path is exercised
    w, U = spla.eigsh(P.A_int, k=k, M=P.B_int, sigma=P.sigma, which="LM", tol=1e-7, maxiter=12000)
    o = np.argsort(w)
    w, U = w[o], U[:, o]
    assert (np.diff(evals) >= 0).all()
    dn = np.abs(np.sqrt(evals) - np.sqrt(w)).max() / P.g.k0
    assert dn < N_EFF_TOL and dn < 1e-10, dn
    V = evecs.cpu().numpy()[:, P.idx].T
    assert np.abs(np.delete(evecs.cpu().numpy(), P.idx, axis=1)).max() == 0.0
    assert _match_fields(V, U, w).max() < FIELD_TOL
    # B-orthonormal like eigsh's output, and true eigenpairs of the pencil
    assert np.abs(V.T @ (P.B_int @ V) - np.eye(k)).max() < 1e-10
    R = P.A_int @ V - (P.B_int @ V) * evals
    assert (np.linalg.norm(R, axis=0) / np.linalg.norm(P.A_int @ V, axis=0)).max() < 1e-8


def test_mode_records_match_oracle(medium, gpu_device):
    P = medium
    solver = TrueVectorialMaxwellSolver(P.g, device=gpu_device)
    modes = solver.solve_vectorial_modes(P.mesh, n_modes_target=10)
    ref = hfield.solve_vectorial_modes(P.g, P.om, n_modes_target=10, fused=True)
    assert len(modes) == len(ref) == 22                            # no truncation to n_modes_target (solver_fem.py:239)
    assert [m["n_eff"] for m in modes] == sorted((m["n_eff"] for m in modes), reverse=True)
    for a, b in zip(modes, ref):
        assert set(a) == set(b)
        assert abs(a["n_eff"] - b["n_eff"]) < N_EFF_TOL and abs(a["beta"] - b["beta"]) < 1e-9
        assert a["Ex_dofs"].shape == b["Ex_dofs"].shape
        for key in ("P_x", "P_y", "confinement", "core_overlap", "div_ratio", "PDL_dB"):
            assert abs(a[key] - b[key]) <= 1e-6 * max(1.0, abs(b[key])), key
        assert a["polarization"] == b["polarization"] and a["is_vectorial"] is True and a["method"] == b["method"]
        assert a.n_eff == a["n_eff"]                               # README-style attribute access
    assert mode_field_errors(modes, ref).max() < FIELD_TOL
    assert solver.last_stats["nconv"] == 22


def test_wavelength_sweep_reuses_analysis_and_is_deterministic(small, gpu_device):
    mesh = small.mesh
    solver = TrueVectorialMaxwellSolver(small.g, device=gpu_device)
    out = {}
    for lam in (1.49, 1.55, 1.60, 1.65, 1.55):
        g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=lam)
        solver.geometry, solver.k0 = g, g.k0
        modes = solver.solve_vectorial_modes(mesh, 6)
        key = (lam, len(out))
        out[key] = modes
    assert len(solver._cache) == 1                                  # one symbolic analysis / context for the sweep
    first, again = out[(1.55, 1)], out[(1.55, 4)]
    for a, b in zip(first, again):                                  # bitwise reproducible (no float atomics)
        assert a["n_eff"] == b["n_eff"]
        np.testing.assert_array_equal(a["Ex_dofs"], b["Ex_dofs"])
    g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.60)
    ref = hfield.solve_vectorial_modes(g, small.om, 6, fused=True)
    got = out[(1.60, 2)]
    assert len(got) == len(ref)
    assert max(abs(a["n_eff"] - b["n_eff"]) for a, b in zip(got, ref)) < N_EFF_TOL


def test_edge_cases_and_errors(gpu_device, built_library):
    # tiny mesh: request clamped to 2 N_solve - 4 (solver_fem.py:196)
    g1 = MCFGeometry(1, 0.0, 0.3, 1.535, 1.0, wavelength_um=1.55)
    sq = unit_square_mesh(2)                                         # 9 interior P2 DOFs -> n = 18
    sq.p[:] = sq.p - 0.5
    solver = TrueVectorialMaxwellSolver(g1, device=gpu_device)
    modes = solver.solve_vectorial_modes(sq, n_modes_target=20)
    st = solver.last_stats
    assert st["n_req"] == 2 * st["N_solve"] - 4 == 14 and st["nconv"] == 14
    ref = hfield.solve_vectorial_modes(g1, MeshTriLite(sq.p, sq.t), 20, fused=True)
    assert len(modes) == len(ref)
    for a, b in zip(modes, ref):
        assert abs(a["n_eff"] - b["n_eff"]) < N_EFF_TOL
    # call-order errors surface as exceptions, never as silent results
    mesh = unit_square_mesh(4)
    sym = _native.Symbolic(mesh.p, mesh.t)
    ctx = _native.Context(sym, gpu_device, max_ncv=45)
    x = ctx.empty(ctx.n2)
    with pytest.raises(RuntimeError):
        ctx.spmv("A", x)                                             # before assemble
    ctx.assemble(_core_table(g1), 2.0, 1.0, 4.0, 1.0)
    with pytest.raises(RuntimeError):
        ctx.solve(x)                                                 # before factor
    with pytest.raises(RuntimeError):
        ctx.lanczos(4, 12, 1e-10, 100, 1.0)                          # before factor
    ctx.factor(1.0)
    with pytest.raises(ValueError):
        ctx.lanczos(4, 60, 1e-10, 100, 1.0)                          # ncv > max_ncv
    with pytest.raises(ValueError):
        ctx.assemble(_core_table(g1), -1.0, 1.0, 4.0, 1.0)           # non-physical permittivity


def test_nineteen_core_layout(gpu_device, built_library):
    g = MCFGeometry(19, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    mesh = generate_mesh(g, 0.35, 0)
    solver = TrueVectorialMaxwellSolver(g, device=gpu_device)
    modes = solver.solve_vectorial_modes(mesh, n_modes_target=8)
    ref = hfield.solve_vectorial_modes(g, MeshTriLite(mesh.p, mesh.t), 8, fused=True)
    assert len(modes) == len(ref) > 0
    assert max(abs(a["n_eff"] - b["n_eff"]) for a, b in zip(modes, ref)) < N_EFF_TOL


def test_north_star_size_properties(c1_geometry, gpu_device, built_library):
    """C1 (N = 90 639, n = 180 742, k = 22): size-independent properties instead of a CPU re-solve."""
    import torch
    g = c1_geometry
    mesh = generate_mesh(g, 1.0, 1)
    sym = _native.Symbolic(mesh.p, mesh.t)
    assert (sym.N, 2 * sym.nsolve) == (90639, 180742)
    ctx = _native.Context(sym, gpu_device, max_ncv=65)
    ctx.assemble(_core_table(g), g.n_core ** 2, g.n_clad ** 2, g.k0, 1.0)
    sigma = shift_estimate(g)
    ctx.factor(sigma)
    evals, V, st = ctx.lanczos(22, 45, 1e-10, 12000, sigma)
    assert st["nconv"] == 22 and (np.diff(evals) >= 0).all()
    n_eff = np.sqrt(evals) / g.k0
    assert (n_eff > g.n_clad).all() and (n_eff < g.n_core).all()
    AV = torch.stack([ctx.spmv("A", V[i]) for i in range(22)])
    BV = torch.stack([ctx.spmv("B", V[i]) for i in range(22)])
    lam = torch.from_numpy(evals).cuda()[:, None]
    res = (AV - lam * BV).norm(dim=1) / AV.norm(dim=1)
    assert res.max().item() < 1e-8                                    # eigen-residuals
    G = V @ BV.T
    assert (G - torch.eye(22, device=G.device, dtype=G.dtype)).abs().max().item() < 1e-10   # B-orthonormal
    # shift-invert consistency: K^-1 (B v) = v / (lambda - sigma)
    x = ctx.solve(BV[3], 0)
    assert ((x - V[3] / (evals[3] - sigma)).norm() / x.norm()).item() < 1e-7
    # n_eff band of SURVEY appendix B at L = 1: 26.122 .. 26.180
    assert abs(evals[0] - 26.122) < 2e-3 and abs(evals[-1] - 26.180) < 2e-3


def test_block_and_single_vector_lanczos_agree(medium, monkeypatch):
    """The block recurrence (4 right-hand sides per pass over the factors) and the single-vector one
    return the same 22 pairs; the dispatch is visible in the stats."""
    P = medium
    P.ctx.factor(P.sigma)
    ev_b, V_b, st_b = P.ctx.lanczos(22, 45, 1e-10, 12000, P.sigma)
    assert st_b["n_block_solves"] > 0 and st_b["n_opinv"] == 4 * st_b["n_block_solves"]
    monkeypatch.setenv("PLFEM_LANCZOS_BLOCK", "0")
    ev_s, V_s, st_s = P.ctx.lanczos(22, 45, 1e-10, 12000, P.sigma)
    assert st_s["n_block_solves"] == 0 and st_s["nconv"] == 22
    assert np.abs(ev_b - ev_s).max() < 1e-9
    Vb = V_b.cpu().numpy()[:, P.idx].T
    Vs = V_s.cpu().numpy()[:, P.idx].T
    assert _match_fields(Vb, Vs, ev_s).max() < FIELD_TOL
    # fewer passes over the factors than single-vector OP applications
    assert st_b["n_block_solves"] < st_s["n_opinv"]


def test_live_kernel_profile_hooks(small):
    P = small
    P.ctx.factor(P.sigma)
    P.ctx.profile_begin(64)
    b = P.embed(np.random.default_rng(2).standard_normal(len(P.idx)))
    for _ in range(4):
        P.ctx.solve(b, 0)
    prof = P.ctx.profile_end()
    T = fe.FrontTree(P.sym)
    tile_levels = sum(1 for lev in range(T.L + 1) if (1 << lev) > 32)
    # whole sweeps and single launches are timed in alternate solves (an event pair around a launch lengthens the sweep it
    # sits in): solves 1 and 3 time the sweeps, solves 2 and 4 the tile-form launches
    assert prof["launches"] == 2 * tile_levels
    assert prof["total_us"] > 0 and prof["bytes"] > 0
    sl = prof["slots"]
    assert sl["fwd_sweep"]["ranges"] == sl["bwd_sweep"]["ranges"] == 2 and sl["spmv_b"]["ranges"] == 0
    assert sl["fwd_sweep"]["bytes"] == sl["bwd_sweep"]["bytes"] >= sl["k_fwd"]["bytes"] > 0
    assert sl["fwd_sweep"]["total_us"] >= sl["k_fwd"]["total_us"] > 0
    # algorithmic bytes of one backward sweep never exceed the bytes of the stored factors + vectors
    assert prof["bytes"] / 3 <= 8.0 * (P.sym.info["solve_entries"] + 4 * P.sym.info["front_doubles"] ** 0.5 * T.nf)


def test_sweep_driver_on_gpu_matches_direct_solves(gpu_device, built_library):
    """Two meshes x two wavelengths through run_sweep (background preparation of the next mesh,
    adopted analysis, context reuse across wavelengths) = the same n_eff as isolated solves."""
    from pl_fem_vectoriel_amd.sweep import SweepItem, run_sweep
    items = []
    for arr in ("linear_2", "triangular_3"):
        for lam in (1.55, 1.60):
            items.append(SweepItem(len(items), arr, 8.0, lam, n_modes=4, mesh_refinement=0.35, mesh_levels=0))
    table, n_local = run_sweep(items, 0, 1, device=gpu_device)
    assert n_local == 4 and sorted(table) == [0, 1, 2, 3]
    for it in items:
        g = it.geometry()
        mesh = generate_mesh(g, it.mesh_refinement, it.mesh_levels)
        direct = TrueVectorialMaxwellSolver(g, device=gpu_device).solve_vectorial_modes(mesh, it.n_modes)
        ref = np.array([m["n_eff"] for m in direct])
        assert len(ref) == len(table[it.index]) and np.abs(ref - table[it.index]).max() < 1e-10


def _fan_mesh(nfan, rings=2):
    """A vertex of degree nfan (centre of a disc triangulated as a fan) plus `rings` rings around it."""
    ang = np.linspace(0.0, 2 * np.pi, nfan, endpoint=False)
    pts = [np.zeros((2, 1))]
    for r in range(1, rings + 1):
        pts.append(r * np.stack([np.cos(ang + 0.5 * r * np.pi / nfan), np.sin(ang + 0.5 * r * np.pi / nfan)]))
    p = np.concatenate(pts, axis=1)
    tris = [[0, 1 + j, 1 + (j + 1) % nfan] for j in range(nfan)]
    for r in range(1, rings):
        a0, b0 = 1 + (r - 1) * nfan, 1 + r * nfan
        for j in range(nfan):
            j1 = (j + 1) % nfan
            tris += [[a0 + j, b0 + j, a0 + j1], [a0 + j1, b0 + j, b0 + j1]]
    return np.ascontiguousarray(p), np.ascontiguousarray(np.array(tris, dtype=np.int32).T)


@pytest.mark.parametrize("nfan", [8, 16, 40, 300])
def test_device_built_csr_pattern_matches_host(nfan, gpu_device, built_library):
    """colind / slot_row are built on the device (k_pattern_fill: per-lane rows, cooperative high-degree rows,
    slow path beyond 256 adjacent elements); the host builds its own copy on demand.  Both must agree."""
    p, t = _fan_mesh(nfan)
    sym = _native.Symbolic(p, t, leaf_elems=8)
    ctx = _native.Context(sym, gpu_device)
    nnz = sym.info["nnz"]
    np.testing.assert_array_equal(ctx.debug_copy("colind", 0, nnz).astype(np.int64), sym.array("colind"))
    np.testing.assert_array_equal(ctx.debug_copy("slot_row", 0, nnz).astype(np.int64), sym.array("slot_row"))
    ctx.close()


def test_device_built_csr_pattern_on_lantern_mesh(small):
    nnz = small.sym.info["nnz"]
    np.testing.assert_array_equal(small.ctx.debug_copy("colind", 0, nnz).astype(np.int64), small.sym.array("colind"))
    np.testing.assert_array_equal(small.ctx.debug_copy("slot_row", 0, nnz).astype(np.int64), small.sym.array("slot_row"))


@pytest.mark.parametrize("arrangement", sorted(__import__("pl_fem_vectoriel_amd.geometry", fromlist=["ARRANGEMENTS"]).ARRANGEMENTS))
def test_every_arrangement_matches_oracle(arrangement, gpu_device, built_library):
    """All 13 cross-section types of geometry_unified.py:74-188 on coarse meshes: same modes as the oracle
    (count after the reference's filters, n_eff, fields) and no pivot perturbation in the factorisation."""
    from pl_fem_vectoriel_amd.geometry import ARRANGEMENTS
    n, variant = ARRANGEMENTS[arrangement]
    g = MCFGeometry(n, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55, variant=variant)
    mesh = generate_mesh(g, 0.35, 0)
    solver = TrueVectorialMaxwellSolver(g, device=gpu_device)
    modes = solver.solve_vectorial_modes(mesh, n_modes_target=6)
    ref = hfield.solve_vectorial_modes(g, MeshTriLite(mesh.p, mesh.t), n_modes_target=6, fused=True)
    assert solver.last_stats["nconv"] == solver.last_stats["n_req"] == 18
    assert solver.last_stats["pivot_perturbations"] == 0
    assert len(modes) == len(ref)
    ne = np.array([m["n_eff"] for m in ref])
    gap = np.full(len(ne), np.inf)
    if len(ne) > 1:
        d = np.abs(np.diff(ne)) / ne[1:]
        gap[:-1] = np.minimum(gap[:-1], d)
        gap[1:] = np.minimum(gap[1:], d)
    for a, b, isolated in zip(modes, ref, gap > 1e-5):
        assert abs(a["n_eff"] - b["n_eff"]) < N_EFF_TOL
        if isolated:      # inside a (near-)degenerate pair the basis, hence every per-mode scalar, is arbitrary
            assert a["polarization"] == b["polarization"]
            for key in ("confinement", "div_ratio", "PDL_dB"):
                assert abs(a[key] - b[key]) <= 1e-5 * max(1.0, abs(b[key])), key
    if modes:
        assert mode_field_errors(modes, ref, rel_gap=1e-5).max() < FIELD_TOL


def test_package_first_import_order_in_a_fresh_process(built_library):
    """Importing the package (and loading libplfem_hip.so) before torch must still end with ONE HIP runtime in the
    process: the loader pulls torch in first (the wheel bundles its own libamdhip64)."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "\n".join([
        "import sys; sys.path.insert(0, %r)" % root,
        "from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh, _native",
        "g = MCFGeometry(2, 8.0, 1.5, 1.535, 1.0)",
        "m = generate_mesh(g, 0.3, 0)",
        "s = _native.Symbolic(m.p, m.t)",
        "c = _native.Context(s, 0)",
        "print('ok', s.N)"])
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().startswith("ok"), out.stderr[-2000:]


def test_readme_form_builds_its_own_mesh(gpu_device, built_library):
    """The documented convenience form (reference README.md:141-160): keyword geometry, `n_modes`, `.solve()` with no
    mesh argument, attribute access on the mode records — same result as the explicit call chain."""
    from pl_fem_vectoriel_amd import PhotonicLanternGeometry
    geom = PhotonicLanternGeometry(arrangement="triangular_3", core_radius_um=1.5, pitch_um=8.0, n_core=1.535,
                                   n_clad=1.0, wavelength_nm=1550.0)
    solver = TrueVectorialMaxwellSolver(geom, n_modes=4, device=gpu_device, mesh_refinement=0.35, mesh_levels=0)
    modes = solver.solve()
    assert len(modes) > 0 and solver.last_stats["n_req"] == 16
    direct = TrueVectorialMaxwellSolver(geom, device=gpu_device).solve_vectorial_modes(generate_mesh(geom, 0.35, 0), 4)
    assert len(direct) == len(modes)
    for a, b in zip(modes, direct):
        assert a.n_eff == a["n_eff"] and a.is_vectorial is True
        assert abs(a.n_eff - b["n_eff"]) < 1e-12


def test_sweep_lanes_give_identical_results(gpu_device, built_library):
    """Two solves in flight per GPU (one host thread + context + stream each) = the same table as one lane."""
    from pl_fem_vectoriel_amd.sweep import SweepItem, run_sweep
    items = []
    for arr in ("linear_2", "triangular_3", "square_2x2_4"):
        for lam in (1.55, 1.60):
            items.append(SweepItem(len(items), arr, 8.0, lam, n_modes=4, mesh_refinement=0.35, mesh_levels=0))
    one, n1 = run_sweep(items, 0, 1, device=gpu_device, lanes=1)
    two, n2 = run_sweep(items, 0, 1, device=gpu_device, lanes=2)
    assert n1 == n2 == 6 and sorted(one) == sorted(two) == list(range(6))
    for i in one:
        np.testing.assert_array_equal(one[i], two[i])



def test_residuals_against_the_assembled_pencil(medium):
    """plfem_residuals = ||A v - lambda B v|| / ||A v|| on the interior pencil, for arbitrary vectors."""
    P = medium
    rng = np.random.default_rng(5)
    V = rng.standard_normal((3, len(P.idx)))
    lam = np.array([26.0, -3.0, 0.5])
    Vd = P.torch.stack([P.embed(v) for v in V])
    got = P.ctx.residuals(lam, Vd)
    for i in range(3):
        av, bv = P.A_int @ V[i], P.B_int @ V[i]
        ref = np.linalg.norm(av - lam[i] * bv) / np.linalg.norm(av)
        assert abs(got[i] - ref) <= 1e-12 * ref


def test_a_posteriori_guard_fires_on_a_bad_factor(small, gpu_device):
    """a factorisation that lost accuracy must never yield silently wrong modes.  The test hook
    perturbs D of the root front after every factorisation; the eigen-solve then converges (its test trusts K^-1),
    the check against the assembled pencil fails, the re-run with refinement inside the operator repairs it."""
    solver = TrueVectorialMaxwellSolver(small.g, device=gpu_device)
    clean = solver.solve_vectorial_modes(small.mesh, 6)
    st = solver.last_stats
    assert st["refined"] is False and st["true_residual"] < 1e-8
    ctx = next(iter(solver._cache.values()))["ctx"]
    ctx.debug_set_perturb(1e-4)
    try:
        modes = solver.solve_vectorial_modes(small.mesh, 6)
        st = solver.last_stats
        assert st["refined"] is True
        assert st["true_residual_first"] > TrueVectorialMaxwellSolver.RESIDUAL_TOL >= st["true_residual"]
        assert len(modes) == len(clean)
        assert max(abs(a["n_eff"] - b["n_eff"]) for a, b in zip(modes, clean)) < 1e-7
        assert mode_field_errors(modes, clean).max() < FIELD_TOL
        # a factor too wrong for one refinement pass to repair: an error (inaccurate factor, or no convergence of the
        # refined operator -- both RuntimeError), never a result
        ctx.debug_set_perturb(0.6)
        with pytest.raises(RuntimeError):
            solver.solve_vectorial_modes(small.mesh, 6)
    finally:
        ctx.debug_set_perturb(0.0)
    again = solver.solve_vectorial_modes(small.mesh, 6)
    assert solver.last_stats["refined"] is False
    for a, b in zip(again, clean):
        assert a["n_eff"] == b["n_eff"]
    with pytest.raises(ValueError):
        ctx.set_option("no_such_option", 1.0)


def test_no_convergence_is_scipys_exception(small):
    """The reference lets scipy.sparse.linalg.ArpackNoConvergence propagate (solver_fem.py:197): ours IS one."""
    P = small
    P.ctx.factor(P.sigma)
    with pytest.raises(spla.ArpackNoConvergence) as ei:
        P.ctx.lanczos(22, 45, 1e-15, 0, P.sigma)       # no restart allowed at an unreachable tolerance
    assert ei.value.eigenvalues is not None and len(ei.value.eigenvalues) == 22


def test_long_basis_many_modes(small, gpu_device):
    """n_modes_target = 80 -> k = 92 pairs, SciPy's ncv = 185 > the 160 columns of round 1: the
    context now holds up to 320 columns, the Ritz rotation stages S in LDS chunk by chunk."""
    solver = TrueVectorialMaxwellSolver(small.g, device=gpu_device)
    solver.solve_vectorial_modes(small.mesh, n_modes_target=80)
    st = solver.last_stats
    assert st["n_req"] == 92 and st["nconv"] == 92 and 185 <= st["ncv"] <= 320
    assert st["true_residual"] < 1e-8
    P = small
    w = spla.eigsh(P.A_int, k=92, M=P.B_int, sigma=P.sigma, which="LM", tol=1e-9, return_eigenvectors=False)
    ctx = next(iter(solver._cache.values()))["ctx"]
    ctx.factor(P.sigma)
    evals, _, _ = ctx.lanczos(92, st["ncv"], 1e-10, 12000, P.sigma)
    assert np.abs(np.sort(w) - evals).max() < 1e-8
    with pytest.raises(ValueError, match="n_modes_target too large"):
        solver.solve_vectorial_modes(small.mesh, n_modes_target=200)


def test_front_too_large_for_the_lds_staging(gpu_device, built_library):
    """The sweeps stage a front's right-hand side in LDS: one front of ~7 000 DOFs exceeds the budget
    for 4 right-hand sides -> the single-vector recurrence is used, results unchanged; one front of ~21 000 DOFs
    exceeds it for one -> plfem_create fails with PLFEM_EINVAL and says why."""
    g1 = MCFGeometry(1, 0.0, 0.3, 1.535, 1.0, wavelength_um=1.55)
    sq = unit_square_mesh(30)
    sq.p[:] = sq.p - 0.5
    solver = TrueVectorialMaxwellSolver(g1, device=gpu_device, leaf_elems=10 ** 6)      # one front = the whole mesh
    modes = solver.solve_vectorial_modes(sq, n_modes_target=6)
    st = solver.last_stats
    sym = next(iter(solver._cache.values()))["sym"]
    assert sym.info["nfronts"] == 1 and sym.info["max_front"] > 5200
    assert st["n_block_solves"] == 0 and st["nconv"] == 18 and st["true_residual"] < 1e-8
    ref = hfield.solve_vectorial_modes(g1, MeshTriLite(sq.p, sq.t), 6, fused=True)
    assert len(modes) == len(ref) and max(abs(a["n_eff"] - b["n_eff"]) for a, b in zip(modes, ref)) < N_EFF_TOL
    solver.clear_cache()
    big = unit_square_mesh(52)
    symb = _native.Symbolic(big.p, big.t, leaf_elems=10 ** 6)
    assert symb.info["max_front"] > 20000
    with pytest.raises(ValueError, match="largest front has .* DOFs.*LDS"):
        _native.Context(symb, gpu_device)


def test_loss_consumer_takes_the_mode_records_of_a_real_solve(medium, gpu_device):
    """Row f2: the reference's documented consumer of the path (losses.py:742-825) fed with the records of a GPU
    solve -- the key contract of row a9 end to end (n_eff, beta, P_x, P_y, PDL_dB, confinement, is_vectorial) -- and
    the same numbers from the oracle's records of the same mesh."""
    from pl_fem_vectoriel_amd.losses import LossCalculator
    P = medium
    solver = TrueVectorialMaxwellSolver(P.g, device=gpu_device)
    modes = solver.solve_vectorial_modes(P.mesh, n_modes_target=10)
    ref_modes = hfield.solve_vectorial_modes(P.g, P.om, n_modes_target=10, fused=True)
    for direction in ("mux", "demux"):
        got = LossCalculator.calculate_physical_losses(modes, P.g, direction, 1550.0)
        ref = LossCalculator.calculate_physical_losses(ref_modes, P.g, direction, 1550.0)
        assert got["success"] and got["is_vectorial"] and got["n_modes_used"] == len(modes) == 22
        for key in ("IL_dB", "MDL_dB", "PDL_dB", "crosstalk_dB", "radiation_loss_dB_per_m", "avg_confinement"):
            assert np.isfinite(got[key])
            assert abs(got[key] - ref[key]) <= 1e-6 * max(1.0, abs(ref[key])), (direction, key)
    assert 0.0 < got["IL_dB"] < 40.0 and -40.0 <= got["crosstalk_dB"] <= -15.0


def test_mesh_generator_feeds_the_solver(gpu_device, built_library):
    """Row f1 end to end: MeshGenerator.generate (reference mesh.py:82-340: point recipe, Delaunay, refinement loop
    driven by SimulationConfig, class-level cache) -> (mesh, basis) -> solve on the GPU = the oracle on that mesh."""
    from pl_fem_vectoriel_amd.mesh import MeshGenerator, SimulationConfig
    MeshGenerator.clear_cache()
    g = MCFGeometry(3, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    cfg = SimulationConfig(mesh_min_points=4000, mesh_target_points=12000)
    mesh, basis = MeshGenerator.generate(g, refinement=0.4, config=cfg)
    assert mesh.nvertices >= 4000 and basis.N == mesh.nvertices + mesh.edges()[0].shape[1]
    again, _ = MeshGenerator.generate(g, refinement=0.4, config=cfg)
    assert again is mesh and MeshGenerator.get_cache_stats()["hits"] == 1
    solver = TrueVectorialMaxwellSolver(g, device=gpu_device)
    modes = solver.solve_vectorial_modes(mesh, n_modes_target=6)
    ref = hfield.solve_vectorial_modes(g, MeshTriLite(mesh.p, mesh.t), n_modes_target=6, fused=True)
    assert len(modes) == len(ref) > 0 and solver.last_stats["N"] == basis.N
    assert max(abs(a["n_eff"] - b["n_eff"]) for a, b in zip(modes, ref)) < N_EFF_TOL
    assert mode_field_errors(modes, ref).max() < FIELD_TOL
    MeshGenerator.clear_cache()
