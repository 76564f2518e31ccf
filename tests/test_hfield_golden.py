"""Committed oracle fixtures of the vectorial path (``tests/golden/hfield_golden.npz``, made by
``tests/golden/make_hfield_golden.py``): the assembled pencil of ``assemble_hfield_system`` (reference
``solver_fem.py:122-169``) and the eigenvalues / n_eff of ``solve_vectorial_modes`` (``:171-239``) on a 7-core and a
2-core mesh small enough to commit.  The CPU test pins the oracle to the fixture (a change of the restatement is a diff
against committed numbers); the GPU tests pin ``plfem_assemble_hfield`` and ``plfem_lanczos_shift_invert`` /
``solve_vectorial_modes`` to the same numbers.  The fixture is oracle output, not reference output: scikit-fem cannot
run here, parity of the assembly half remains unpinned."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import hfield
from oracle.p2 import MeshTriLite
from pl_fem_vectoriel_amd import MCFGeometry
from pl_fem_vectoriel_amd.geometry import ARRANGEMENTS
from pl_fem_vectoriel_amd.mesh import TriMesh

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hfield_golden.npz")
CASES = ("hex7", "lin2")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def _case(Z, name):
    pitch, r, n_core, n_clad, lam, n_modes = Z[f"{name}_params"]
    n, variant = ARRANGEMENTS[str(Z[f"{name}_arrangement"])]
    g = MCFGeometry(n, float(pitch), float(r), float(n_core), float(n_clad), wavelength_um=float(lam), variant=variant)
    return g, Z[f"{name}_p"], Z[f"{name}_t"], int(n_modes)


def _csr(Z, name, key, n):
    return sp.csr_matrix((Z[f"{name}_{key}_data"], Z[f"{name}_{key}_indices"], Z[f"{name}_{key}_indptr"]), shape=(n, n))


def _same(M, G, rel):
    assert M.shape == G.shape
    assert abs(M - G).max() <= rel * abs(G).max()


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_the_committed_pencil_and_eigenvalues(golden, name):
    Z = golden
    g, p, t, n_modes = _case(Z, name)
    mesh = MeshTriLite(p, t)
    for asm in (hfield.assemble_hfield_system, hfield.assemble_hfield_system_fused):
        A, B, basis, Dxx, Dyy, Dxy, Minv = asm(g, mesh)
        N = basis.N
        _same(A, _csr(Z, name, "A", 2 * N), 1e-13)
        _same(B, _csr(Z, name, "B", 2 * N), 1e-13)
        for key, M in (("Dxx", Dxx), ("Dxy", Dxy), ("Dyy", Dyy), ("Minv", Minv)):
            _same(M, _csr(Z, name, key, N), 1e-13)
        np.testing.assert_array_equal(np.asarray(basis.get_dofs().all()), Z[f"{name}_boundary"])
    A, B, *_ = hfield.assemble_hfield_system(g, mesh)
    G = _csr(Z, name, "A", A.shape[0])
    G.eliminate_zeros()
    A = A.tocsr()
    A.sort_indices()
    np.testing.assert_array_equal(A.indptr, G.indptr)           # the pattern too (explicit zeros dropped, appendix A7)
    np.testing.assert_array_equal(A.indices, G.indices)
    assert abs(hfield.shift_estimate(g) - float(Z[f"{name}_sigma"])) == 0.0
    modes, raw = hfield.solve_vectorial_modes(g, mesh, n_modes, fused=True, return_raw=True)
    np.testing.assert_allclose(np.sort(raw["beta_sq"]), Z[f"{name}_beta_sq"], rtol=1e-11, atol=0)
    assert len(modes) == len(Z[f"{name}_n_eff"])
    np.testing.assert_allclose([m["n_eff"] for m in modes], Z[f"{name}_n_eff"], rtol=1e-12, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_assembly_matches_the_committed_pencil(golden, name, gpu_device, built_library):
    from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver
    Z = golden
    g, p, t, _n = _case(Z, name)
    solver = TrueVectorialMaxwellSolver(g, device=gpu_device)
    A, B, basis, Dxx, Dyy, Dxy, Minv = solver.assemble_hfield_system(TriMesh(p, t))
    N = basis.N
    _same(A, _csr(Z, name, "A", 2 * N), 1e-12)
    _same(B, _csr(Z, name, "B", 2 * N), 1e-13)
    for key, M in (("Dxx", Dxx), ("Dxy", Dxy), ("Dyy", Dyy), ("Minv", Minv)):
        _same(M, _csr(Z, name, key, N), 1e-12)
    np.testing.assert_array_equal(np.asarray(basis.get_dofs().all()), Z[f"{name}_boundary"])
    solver.clear_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_eigenvalues_match_the_committed_ones(golden, name, gpu_device, built_library):
    from pl_fem_vectoriel_amd import _native
    from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver, _core_table
    Z = golden
    g, p, t, n_modes = _case(Z, name)
    mesh = TriMesh(p, t)
    # the C-ABI calls themselves: plfem_assemble_hfield -> plfem_factor -> plfem_lanczos_shift_invert
    sym = _native.Symbolic(mesh.p, mesh.t)
    ctx = _native.Context(sym, gpu_device, max_ncv=65)
    ctx.assemble(_core_table(g), g.n_core ** 2, g.n_clad ** 2, g.k0, 1.0)
    sigma = float(Z[f"{name}_sigma"])
    ctx.factor(sigma)
    k = len(Z[f"{name}_beta_sq"])
    evals, _evecs, st = ctx.lanczos(k, 45, 1e-10, 12000, sigma)
    assert st["nconv"] == k and ctx.timings()["pivot_perturbations"] == 0
    np.testing.assert_allclose(np.sort(evals), Z[f"{name}_beta_sq"], rtol=1e-10, atol=0)
    ctx.close()
    # and through the reference's class surface
    solver = TrueVectorialMaxwellSolver(g, device=gpu_device)
    modes = solver.solve_vectorial_modes(mesh, n_modes)
    assert len(modes) == len(Z[f"{name}_n_eff"])
    np.testing.assert_allclose([m["n_eff"] for m in modes], Z[f"{name}_n_eff"], rtol=1e-10, atol=0)
    solver.clear_cache()
