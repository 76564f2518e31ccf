"""Pivoting of the block LDL^T: the reference factorises A - sigma B with SuperLU's partial pivoting
(solver_fem.py:197 -> scipy arpack.py:915) and returns modes for every mesh; this path pivots node pair by node pair in
a static order (scalar or 2 x 2 pivots, kernels_front.hip) and must do the same without its a-posteriori guard stepping
in.  Random coarse cross-sections -- every layout of geometry_unified.py:98-184, pitch 6-10 um, wavelength
1.45-1.65 um, mesh recipe refinement 0.3-0.6 (the recipe's hull slivers included) -- each solved cold and compared with
the oracle: no perturbed pivot, no refined re-run, first-pass eigen-residual below 1e-8, the 18 eigenvalues nearest the
shift equal to the oracle's.

The list has 104 vectorial + 24 scalar cases (``PLFEM_PIVOT_CASES=104 PLFEM_PIVOT_CASES_SCALAR=24``: what every change
of the factorisation was checked with in round 3, last full run recorded in profiles/r03_pivoting_full.txt); a plain
``-m gpu`` run takes the first 48 + 12 of them -- most of a case's time is the oracle's SuperLU + ARPACK on the host, and
the whole GPU suite has to fit the driver's time limit on a busy box."""
import os

import numpy as np
import pytest

from oracle import hfield
from oracle.p2 import MeshTriLite
from pl_fem_vectoriel_amd import MCFGeometry
from pl_fem_vectoriel_amd.geometry import ARRANGEMENTS
from pl_fem_vectoriel_amd.mesh import generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver

pytestmark = pytest.mark.gpu

N_ALL = 104                                                      # the fixed list (seeded): a prefix is run by default
N_CASES = max(4, min(N_ALL, int(os.environ.get("PLFEM_PIVOT_CASES", "48"))))
N_CASES_SCALAR = max(1, min(24, int(os.environ.get("PLFEM_PIVOT_CASES_SCALAR", "12"))))


def random_cross_sections(n=N_ALL, seed=20261004):
    rng = np.random.default_rng(seed)
    names = sorted(a for a in ARRANGEMENTS if a != "single_1")
    for t in range(n):
        arr = names[int(rng.integers(len(names)))]
        yield t, arr, float(rng.uniform(6.0, 10.0)), float(rng.uniform(1.45, 1.65)), float(rng.uniform(0.3, 0.6))


# Cases whose eigenvalues differ from eigsh run with the REFERENCE's own arguments (k = 18, tol 1e-7) and agree with a wider,
# tighter eigsh instead -- i.e. where this path returns a different set than the reference's call would.  The list is
# asserted, so that a new divergence from the reference-argument run shows up as a failure, not as a silent switch of
# oracles.  All four hold an exactly degenerate pair at the far edge of the wanted set (see below): ARPACK's
# single-vector recurrence finds the second copy of such a pair out of rounding noise only, so whether a run with the
# reference's arguments returns it depends on the start vector and on the BLAS's rounding -- with the per-case start vector
# below, 38 and 44 of the default 48 cases and 95, 96 of the full list of 104 take the fallback (round 4,
# profiles/r04_pivoting_full.txt); with ARPACK's own start vector it was 38 alone.  The assertion is "no case outside this
# list": a listed case that happens to agree with the reference-argument run is fine.
KNOWN_WIDE_ORACLE_CASES = {38, 44, 95, 96}


@pytest.mark.parametrize("chunk", range(4))          # (a quarter of the cases each -- 12 by default, 26 of the full list: a test that stays silent for minutes looks hung)
def test_random_coarse_cross_sections_factor_cleanly_and_match_the_oracle(chunk, gpu_device, built_library):
    from scipy.sparse.linalg import eigsh
    worst_dn, worst_res = 0.0, 0.0
    took_wide_oracle = set()
    for t, arr, pitch, lam, refinement in list(random_cross_sections())[:N_CASES][chunk::4]:
        n, variant = ARRANGEMENTS[arr]
        g = MCFGeometry(n, pitch, 1.5, 1.535, 1.0, wavelength_um=lam, variant=variant)
        mesh = generate_mesh(g, refinement, 0)
        solver = TrueVectorialMaxwellSolver(g, device=gpu_device)
        modes = solver.solve_vectorial_modes(mesh, 6)
        st = solver.last_stats
        case = (t, arr, round(pitch, 3), round(lam, 4), round(refinement, 3))
        assert st["pivot_perturbations"] == 0 and st["refined"] is False, (case, st["pivot_perturbations"], st["true_residual_first"])
        assert st["true_residual"] < 1e-8, (case, st["true_residual"])
        # the oracle's pencil and eigsh with the reference's arguments (k = n_req = 18, tol 1e-7; eigenvalues only: the
        # comparison below is on the n_req eigenvalues nearest sigma themselves, before the reference's filters -- for an
        # exactly degenerate pair the div_ratio of a member depends on the basis eigsh happens to return inside the pair,
        # so a pair that straddles the filter's threshold is filtered differently by ANY two runs, the reference's own
        # included)
        A, B, basis, *_ = hfield.assemble_hfield_system_fused(g, MeshTriLite(mesh.p, mesh.t))
        A_int, B_int, _interior = hfield.restrict_interior(A, B, basis)
        sigma = hfield.shift_estimate(g)
        assert abs(sigma - st["sigma"]) <= 1e-12 * abs(sigma)
        # (a start vector of the case's own: without one ARPACK draws from a generator whose state depends on every earlier
        # eigsh call of the process, and which cases end up in KNOWN_WIDE_ORACLE_CASES would depend on the test selection)
        v0 = np.random.default_rng(1000 + t).standard_normal(A_int.shape[0])
        raw = {"A_int": A_int, "B_int": B_int, "sigma": sigma,
               "beta_sq": eigsh(A_int, k=18, M=B_int, sigma=sigma, which="LM", tol=1e-7, maxiter=12000, return_eigenvectors=False,
                                v0=v0)}
        got = np.sort(np.sqrt(st["beta_sq"])) / g.k0
        want = np.sort(np.sqrt(raw["beta_sq"])) / g.k0
        if np.abs(got - want).max() >= 1e-9:
            # The comparison is with what eigsh's contract promises -- the 18 eigenvalues nearest sigma -- taken from a
            # wider and tighter run (k = 26, tol 1e-10): with the reference's own arguments (k = 18, tol 1e-7) ARPACK's
            # single-vector recurrence now and then returns only ONE copy of an exactly degenerate pair at the far edge of
            # the wanted set and the next eigenvalue in its place (case 38 of this list: pair at sigma - 0.36247, eigsh
            # returns one copy and sigma - 0.36298); the block recurrence of the HIP path finds both copies.
            wide = eigsh(raw["A_int"], k=26, M=raw["B_int"], sigma=raw["sigma"], which="LM", tol=1e-10, maxiter=12000,
                         return_eigenvectors=False, v0=v0)
            wide = wide[np.argsort(np.abs(wide - raw["sigma"]))][:18]
            want = np.sort(np.sqrt(wide)) / g.k0
            took_wide_oracle.add(t)
        assert len(got) == len(want) == 18
        dn = float(np.abs(got - want).max())
        assert dn < 1e-9, (case, dn)                      # (north_star's bar is 5e-5)
        assert 0 < len(modes) <= 18, (case, len(modes))
        worst_dn, worst_res = max(worst_dn, dn), max(worst_res, st["true_residual"])
        solver.clear_cache()
    print(f"chunk {chunk}: max |dn_eff| {worst_dn:.2e}, max first-pass residual {worst_res:.2e}, wide oracle for {sorted(took_wide_oracle)}")
    assert took_wide_oracle <= KNOWN_WIDE_ORACLE_CASES, sorted(took_wide_oracle - KNOWN_WIDE_ORACLE_CASES)


def test_random_cross_sections_of_the_scalar_pencil(gpu_device, built_library):
    """The same for ``ScalarHelmholtzSolver`` (one unknown per node: a pivot pair is two neighbouring NODES, and two nodes
    of one hull sliver make a pair with condition 1e9 -- the case that rules out 2 x 2 pivots through the explicit inverse
    everywhere): 24 random cross-sections, raw eigenvalues against the oracle's."""
    from scipy.sparse.linalg import eigsh
    from oracle import scalar
    from pl_fem_vectoriel_amd.solver_fem import ScalarHelmholtzSolver
    worst = 0.0
    for t, arr, pitch, lam, refinement in list(random_cross_sections(24, seed=7))[:N_CASES_SCALAR]:
        n, variant = ARRANGEMENTS[arr]
        g = MCFGeometry(n, pitch, 1.5, 1.535, 1.0, wavelength_um=lam, variant=variant)
        mesh = generate_mesh(g, refinement, 0)
        solver = ScalarHelmholtzSolver(g, device=gpu_device)
        modes = solver.solve(mesh, n_modes_target=6)
        st = solver.last_stats
        case = (t, arr, round(pitch, 3), round(lam, 4), round(refinement, 3))
        assert st["pivot_perturbations"] == 0 and st["refined"] is False, (case, st["pivot_perturbations"], st["true_residual_first"])
        assert st["true_residual"] < ScalarHelmholtzSolver.RESIDUAL_TOL, (case, st["true_residual"])
        ref, raw = scalar.solve(g, MeshTriLite(mesh.p, mesh.t), 6, return_raw=True)
        # the wanted set (k = 14 nearest sigma) from a wider, tighter eigsh: see the vectorial test
        wide = eigsh(raw["A"], k=20, M=raw["M"], sigma=raw["sigma"], which="LM", tol=1e-10, maxiter=6000, return_eigenvectors=False)
        wide = wide[np.argsort(np.abs(wide - raw["sigma"]))][:14]
        ne_want = np.sort(np.sqrt(-wide[wide < 0])) / g.k0
        ne_want = ne_want[(ne_want > g.n_clad) & (ne_want < g.n_core * 1.005)][::-1]
        got = np.array([m["n_eff"] for m in modes])
        assert len(got) == len(ne_want) > 0, (case, len(got), len(ne_want))
        dn = float(np.abs(got - ne_want).max())
        assert dn < 1e-9, (case, dn)
        worst = max(worst, dn)
        solver.clear_cache()
    print(f"scalar pencil, {N_CASES_SCALAR} cross-sections: max |dn_eff| {worst:.2e}")



def _first_pair_singular_sigma(sym, A, B, which=0):
    """sigma at which the FIRST node pair a leaf front eliminates is singular as a whole: an eigenvalue of the 2 x 2 pencil
    (A_pp, B_pp) of that node's (Hx, Hy) DOFs (nothing is eliminated before it, so its Schur complement is K_pp itself)."""
    N = sym.N
    fs_true, fptr, fnodes = sym.array("fs_true"), sym.array("fnode_ptr"), sym.array("fnodes")
    nf = len(fs_true)
    leaf0 = (nf + 1) // 2 - 1
    f = next(q for q in range(leaf0, nf) if fs_true[q] > 0)
    node = int(fnodes[fptr[f]])
    idx = [node, N + node]
    w = np.sort(np.linalg.eigvals(np.linalg.solve(B[idx][:, idx].toarray(), A[idx][:, idx].toarray())).real)
    return float(w[which]), node, f


@pytest.mark.parametrize("which", [0, 1])
def test_a_pair_singular_as_a_whole_yields_the_oracles_modes(which, gpu_device, built_library):
    """The one situation where this path and the reference differ in KIND.  The reference factorises with
    SuperLU's partial pivoting (solver_fem.py:197 -> scipy arpack.py:915) and returns modes for any regular A - sigma B; the
    block LDL^T here pivots inside node pairs in a static order.  sigma is chosen so that the first pair a leaf front
    eliminates is singular as a whole -- A - sigma B itself is perfectly regular, eigsh does not even notice -- which no
    choice inside the pair can repair.  Required: the vanishing pivot is replaced and COUNTED, the a-posteriori policy of
    plfem_solve_modes repeats the eigen-solve with refinement inside the operator, and the modes are the oracle's -- not
    an exception.  (Emulated on the CPU first: scripts/singular_pair_emulation.py.)"""
    from scipy.sparse.linalg import eigsh
    from pl_fem_vectoriel_amd import _native
    from pl_fem_vectoriel_amd.solver_fem import _core_table
    g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    mesh = generate_mesh(g, 0.3, 0)
    A, B, basis, *_ = hfield.assemble_hfield_system_fused(g, MeshTriLite(mesh.p, mesh.t))
    A_int, B_int, interior = hfield.restrict_interior(A, B, basis)
    sym = _native.Symbolic(mesh.p, mesh.t)
    sigma, node, front = _first_pair_singular_sigma(sym, A.tocsr(), B.tocsr(), which)
    k, ncv = 8, 64
    want = eigsh(A_int, k=k, M=B_int, sigma=sigma, which="LM", tol=1e-12, maxiter=12000, return_eigenvectors=False)
    want = np.sort(want)
    ctx = _native.Context(sym, gpu_device, max_ncv=ncv + 8)
    import torch
    host = torch.empty((k, 2 * sym.nsolve), dtype=torch.float64, pin_memory=True)
    evals, post, frac, resid, st = ctx.solve_modes(_core_table(g), g.n_core ** 2, g.n_clad ** 2, g.k0, 1.0, sigma, k, ncv, 1e-8,
                                                   12000, 1e-7, 1e-10, modes_host=host)
    print(f"sigma* = {sigma!r} (node {node}, front {front}): perturbed {st['pivot_perturbations']}, refined {st['refined']}, "
          f"residual first {st['true_residual_first']:.2e} final {st['true_residual']:.2e}, OP applications {st['n_opinv']}")
    assert st["pivot_perturbations"] >= 1                      # the pair was seen to vanish, and said so
    assert st["refined"] is True                               # ... which alone sends the solve through the refined pass
    assert st["true_residual"] <= 1e-7 and float(resid.max()) <= 1e-7
    assert np.abs(np.sort(evals) - want).max() <= 1e-9 * np.abs(want).max(), (np.sort(evals), want)
    # the vectors: eigen-residual against the ORACLE's pencil, interior parts as delivered to the host
    V = host.numpy()
    for i in range(k):
        v = V[i]
        r = A_int @ v - evals[i] * (B_int @ v)
        assert np.linalg.norm(r) <= 1e-6 * np.linalg.norm(A_int @ v), (i, np.linalg.norm(r) / np.linalg.norm(A_int @ v))
    ctx.close()
