"""CPU tests of the product library's host half: the C-ABI loads and exports every declared symbol,
the symbolic analysis reproduces the oracle's (scikit-fem compatible) numbering and sparsity, and the
nested-dissection front tree is a valid elimination structure (checked by a NumPy emulation of the
multifrontal LDL^T against SciPy's splu)."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import front_emulation as fe
from oracle import hfield
from oracle.p2 import MeshTriLite, P2Basis
from pl_fem_vectoriel_amd import _native
from pl_fem_vectoriel_amd.mesh import TriMesh, generate_mesh, unit_square_mesh
from pl_fem_vectoriel_amd.solver_fem import shift_estimate

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built_library):
    header = open(os.path.join(ROOT, "include", "plfem.h")).read()
    # the test hooks are declared under PLFEM_TEST_HOOKS and belong to the add-on library, not to the product
    product, hooks = re.split(r"#ifdef PLFEM_TEST_HOOKS", header)
    hooks = hooks.split("#endif /* PLFEM_TEST_HOOKS */")[0]
    declared = set(re.findall(r"\b(plfem_[a-z_]+)\s*\(", product))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    for name in declared:
        assert hasattr(built_library, name), name
    declared_hooks = set(re.findall(r"\b(plfem_[a-z_]+)\s*\(", hooks))
    assert declared_hooks == set(_native.TEST_HOOK_EXPORTS), declared_hooks ^ set(_native.TEST_HOOK_EXPORTS)


def test_product_library_exports_no_test_hook(built_library):
    """A shipped libplfem_hip.so has no entry point (and no option) that makes results wrong: the plfem_debug_* hooks
    live in libplfem_testhooks.so only."""
    for name in _native.TEST_HOOK_EXPORTS:
        assert not hasattr(built_library, name), name
    hooks = _native.load_test_hooks()
    for name in _native.TEST_HOOK_EXPORTS:
        assert hasattr(hooks, name), name
    syms = os.popen(f"nm -D --defined-only {_native.LIB_PATH}").read()
    assert "plfem_debug" not in syms
    assert b"debug_perturb" not in open(_native.LIB_PATH, "rb").read()


def test_no_device_means_loud_failure(built_library, c1_geometry):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    mesh = unit_square_mesh(3)
    sym = _native.Symbolic(mesh.p, mesh.t)
    with pytest.raises(RuntimeError):
        _native.Context(sym)
    # the C entry point itself also refuses (no silent CPU path)
    h = ctypes.c_void_p()
    err = ctypes.create_string_buffer(256)
    rc = built_library.plfem_create(sym._h, 0, None, 45, None, 0, ctypes.byref(h), err, 256)
    assert rc == _native.PLFEM_EHIP and b"no HIP device" in err.value


@pytest.mark.parametrize("which", ["square", "lantern", "shuffled"])
def test_numbering_and_pattern_match_oracle(built_library, c1_geometry, which):
    if which == "square":
        mesh = unit_square_mesh(6)
    else:
        mesh = generate_mesh(c1_geometry, 0.4, 1 if which == "lantern" else 0)
    p, t = mesh.p, mesh.t.copy()
    if which == "shuffled":                    # unsorted columns / permuted elements: sort_t must normalise
        rng = np.random.default_rng(3)
        t = t[:, rng.permutation(t.shape[1])]
        for e in range(t.shape[1]):
            t[:, e] = t[rng.permutation(3), e]
    sym = _native.Symbolic(p, t, leaf_elems=16, nthreads=2)
    b = P2Basis(MeshTriLite(p, t))
    assert sym.N == b.N and sym.nedges == b.nedges
    np.testing.assert_array_equal(sym.array("edof").reshape(6, -1), b.element_dofs)
    np.testing.assert_array_equal(sym.array("doflocs").reshape(2, -1), b.doflocs)
    np.testing.assert_array_equal(np.nonzero(sym.array("bmask"))[0], b.get_dofs().all())
    interior = np.setdiff1d(np.arange(b.N), b.get_dofs().all())
    np.testing.assert_array_equal(sym.array("interior"), interior)
    # structural CSR pattern = pattern of sum_e |P_e| (before explicit-zero elimination)
    ed = b.element_dofs
    rows = np.broadcast_to(ed.T[:, :, None], (ed.shape[1], 6, 6)).ravel()
    cols = np.broadcast_to(ed.T[:, None, :], (ed.shape[1], 6, 6)).ravel()
    P = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(b.N, b.N)).tocsr()
    P.sort_indices()
    np.testing.assert_array_equal(sym.array("rowptr"), P.indptr)
    np.testing.assert_array_equal(sym.array("colind"), P.indices)
    np.testing.assert_array_equal(sym.array("slot_row"), np.repeat(np.arange(b.N), np.diff(P.indptr)))
    # node -> element adjacency the device gather walks: every (element, local node) exactly once, filed
    # under its node, elements ascending (=> deterministic summation order)
    nptr, nadj, nloc = sym.array("nptr"), sym.array("nadj"), sym.array("nloc")
    assert nptr[0] == 0 and nptr[-1] == 6 * ed.shape[1] == len(nadj) == len(nloc)
    node_of = np.repeat(np.arange(b.N), np.diff(nptr))
    np.testing.assert_array_equal(ed[nloc, nadj], node_of)
    assert len(set(zip(nadj.tolist(), nloc.tolist()))) == 6 * ed.shape[1]
    for i in np.random.default_rng(0).integers(0, b.N, 200):
        assert (np.diff(nadj[nptr[i]:nptr[i + 1]]) > 0).all()


def test_front_tree_invariants(built_library, c1_geometry):
    mesh = generate_mesh(c1_geometry, 0.5, 0)
    sym = _native.Symbolic(mesh.p, mesh.t, leaf_elems=24)
    T = fe.FrontTree(sym)
    owner = sym.array("owner")
    bmask = sym.array("bmask").astype(bool)
    assert (owner[bmask] == -1).all() and (owner[~bmask] >= 0).all()
    seen = np.zeros(sym.N, dtype=int)
    for f in range(T.nf):
        fn = T.nodes(f)
        own = fn[:T.fs[f]]
        own = own[own >= 0]
        assert (owner[own] == f).all()
        seen[own] += 1
        bnd = fn[T.fs[f]:]
        bnd = bnd[bnd >= 0]
        # boundary nodes are owned by proper ancestors
        for v in bnd[:: max(1, len(bnd) // 8)]:
            a = owner[v]
            g = f
            while g > a:
                g = (g - 1) // 2
            assert g == a and a != f
        assert T.fs[f] % 8 == 0 and T.fb[f] % 8 == 0
    assert (seen[~bmask] == 1).all() and (seen[bmask] == 0).all()
    assert T.fb[0] == 0                                            # the root has no boundary
    # every element sits in exactly one leaf, with all its non-Dirichlet nodes present in that front
    epos = T.epos
    edof = sym.array("edof").reshape(6, -1)
    leaf = sym.array("leaf_of_elem")
    for e in np.random.default_rng(1).integers(0, sym.ne, 300):
        fn = T.nodes(T.leaf0 + leaf[e])
        for a in range(6):
            if bmask[edof[a, e]]:
                assert epos[a, e] == -1
            else:
                assert fn[epos[a, e]] == edof[a, e]


@pytest.mark.parametrize("refinement,leaf", [(0.35, 12), (0.5, 24)])
def test_multifrontal_emulation_matches_splu(built_library, c1_geometry, refinement, leaf):
    """The front tree + unpivoted block LDL^T with inverted L11 (what the HIP kernels execute) solves
    (A - sigma B) x = b to the accuracy of SuperLU on a mesh that contains near-degenerate triangles."""
    g = c1_geometry
    mesh = generate_mesh(g, refinement, 0)
    sym = _native.Symbolic(mesh.p, mesh.t, leaf_elems=leaf)
    om = MeshTriLite(mesh.p, mesh.t)
    basis = P2Basis(om)
    sigma = shift_estimate(g)
    Ke = fe.element_K(hfield.element_matrices(g, basis), g.k0 ** 2, sigma)
    T = fe.FrontTree(sym)
    Fs, Ds = fe.factor(T, Ke)
    A, B, _, _, _, _, _ = hfield.assemble_hfield_system_fused(g, om)
    A_int, B_int, interior = hfield.restrict_interior(A, B, basis)
    N = sym.N
    idx = np.concatenate([interior, interior + N])
    K = (A_int - sigma * B_int).tocsc()
    rhs = np.zeros(2 * N)
    rhs[idx] = np.random.default_rng(0).standard_normal(len(idx))
    x = fe.solve(T, Fs, Ds, rhs)
    xs = spla.splu(K).solve(rhs[idx])
    assert np.abs(np.delete(x, idx)).max() == 0.0
    assert np.linalg.norm(x[idx] - xs) / np.linalg.norm(xs) < 1e-8


def test_malformed_meshes_are_rejected(built_library):
    p = np.array([[0, 1, 0, 1], [0, 0, 1, 1.0]])
    with pytest.raises(ValueError):
        _native.Symbolic(p, np.array([[0], [1], [7]]))            # vertex out of range
    with pytest.raises(ValueError):
        _native.Symbolic(p, np.array([[0], [1], [1]]))            # repeated vertex
    with pytest.raises(ValueError):
        _native.Symbolic(p[:, :2], np.zeros((3, 0), dtype=np.int32))   # empty
    with pytest.raises(ValueError):                                # non-manifold: three triangles on one edge
        _native.Symbolic(np.array([[0, 1, 0, 1, .5], [0, 0, 1, 1, -1.0]]),
                         np.array([[0, 0, 0], [1, 1, 1], [2, 3, 4]]))
    with pytest.raises(ValueError):                                # single triangle: every DOF is on the boundary
        _native.Symbolic(p[:, :3], np.array([[0], [1], [2]]))
    with pytest.raises(ValueError):
        TriMesh(p, np.array([[0, 1], [1, 2]]))


def test_closed_form_row_lengths_on_irregular_topology(built_library):
    """rowptr comes from closed-form row lengths (1 + 2a + d per vertex, 3 + 3m per edge node); they must equal
    the lengths of the sorted unique column lists also where the element fan around a vertex is not a disc:
    a bow-tie vertex, a mesh with a hole, isolated strips."""
    import scipy.sparse as sp
    from scipy.spatial import Delaunay
    rng = np.random.default_rng(11)
    pts = rng.random((2, 400))
    tri = Delaunay(pts.T).simplices.T.astype(np.int32)
    cen = pts[:, tri].mean(axis=1)
    keep = (np.hypot(cen[0] - 0.5, cen[1] - 0.5) > 0.18) & (rng.random(tri.shape[1]) > 0.25)   # hole + random removals
    cases = [(pts, tri[:, keep])]
    # bow tie: two triangles sharing only vertex 2, plus a fan attached to one of them
    p = np.array([[0.0, 1.0, 0.5, 0.0, 1.0, 1.5], [0.0, 0.0, 0.5, 1.0, 1.0, 0.2]])
    t = np.array([[0, 1, 2], [2, 3, 4], [1, 5, 2]], dtype=np.int32).T
    cases.append((p, t))
    for p, t in cases:
        used = np.unique(t)
        remap = -np.ones(p.shape[1], dtype=np.int32); remap[used] = np.arange(len(used), dtype=np.int32)
        p, t = np.ascontiguousarray(p[:, used]), remap[t]
        sym = _native.Symbolic(p, t, leaf_elems=4, nthreads=3)
        ed = sym.array("edof").reshape(6, -1)
        rows = np.broadcast_to(ed.T[:, :, None], (ed.shape[1], 6, 6)).ravel()
        cols = np.broadcast_to(ed.T[:, None, :], (ed.shape[1], 6, 6)).ravel()
        P = sp.coo_matrix((np.ones(rows.size), (rows, cols)), shape=(sym.N, sym.N)).tocsr()
        P.sort_indices()
        np.testing.assert_array_equal(sym.array("rowptr"), P.indptr)
        np.testing.assert_array_equal(sym.array("colind"), P.indices)
        assert sym.info["nnz"] == P.nnz


def test_analysis_is_identical_for_every_thread_count(built_library, c1_geometry):
    """The tree runs on the pool while the numbering / adjacency chain runs on a side thread (nthreads > 1): every
    array the device code or the tests can see must come out the same as from the sequential single-thread build."""
    mesh = generate_mesh(c1_geometry, 0.6, 1)
    ref = _native.Symbolic(mesh.p, mesh.t, nthreads=1)
    names = ("edof", "tsorted", "edges", "doflocs", "bmask", "interior", "int_index", "rowptr", "colind", "nptr", "nadj",
             "nloc", "leaf_of_elem", "leaf_elem_ptr", "leaf_elems", "owner", "fs", "fb", "fnode_ptr", "fnodes", "cinv0",
             "cinv1", "foff", "prow", "npos")
    for nt in (2, 5, 8, 16):
        sym = _native.Symbolic(mesh.p, mesh.t, nthreads=nt)
        for key in ("N", "nsolve", "nnz", "levels", "nfronts", "front_doubles", "max_front", "solve_entries", "factor_flops"):
            assert sym.info[key] == ref.info[key], (nt, key)
        for name in names:
            np.testing.assert_array_equal(sym.array(name), ref.array(name), err_msg=f"{name} with {nt} threads")


def test_malformed_mesh_is_reported_from_the_side_chain(built_library):
    """A non-manifold mesh is detected by the edge numbering, which now runs beside the tree: the error must still come
    back (and the tree thread must have been joined: no crash, no hang) for every thread count."""
    p = np.array([[0.0, 1.0, 0.0, 1.0, 0.5], [0.0, 0.0, 1.0, 1.0, -1.0]])
    t = np.array([[0, 1, 0, 0], [1, 3, 1, 1], [2, 2, 4, 3]], dtype=np.int32)      # edge (0, 1) in three triangles
    for nt in (1, 4):
        with pytest.raises(ValueError, match="non-manifold"):
            _native.Symbolic(p, t, nthreads=nt)


def test_allocator_tuning_is_opt_in(monkeypatch):
    """Loading the library must not change the host process's malloc policy unless the application asks for it."""
    calls = []

    class FakeLibc:
        def mallopt(self, a, b):
            calls.append((a, b))
            return 1

    monkeypatch.setattr(_native.ctypes, "CDLL", lambda name: FakeLibc())
    monkeypatch.delenv("PLFEM_MALLOC_TUNE", raising=False)
    _native._tune_host_allocator()
    monkeypatch.setenv("PLFEM_MALLOC_TUNE", "0")
    _native._tune_host_allocator()
    assert calls == []
    monkeypatch.setenv("PLFEM_MALLOC_TUNE", "1")
    _native._tune_host_allocator()
    assert calls == [(-3, 32 << 20), (-1, 512 << 20)]      # M_MMAP_THRESHOLD, M_TRIM_THRESHOLD


def test_analysis_cache_key_follows_the_mesh_content():
    """solver_fem.mesh_key: an in-place edit of mesh.p (same object, same shapes) must change the key, an equal copy
    must not (the cache used to be keyed on id(mesh))."""
    from pl_fem_vectoriel_amd.solver_fem import mesh_key
    mesh = unit_square_mesh(5)
    k0 = mesh_key(mesh)
    same = TriMesh(mesh.p.copy(), mesh.t.copy())
    assert mesh_key(same) == k0
    mesh.p[0, 3] += 1e-9                     # in place
    assert mesh_key(mesh) != k0
    mesh.p[0, 3] -= 1e-9
    t = mesh.t.copy()
    t[:, [0, 1]] = t[:, [1, 0]]
    assert mesh_key(TriMesh(mesh.p, t)) != k0
    # Fortran-ordered / strided views of the same numbers hash like the contiguous arrays
    assert mesh_key(TriMesh(np.asfortranarray(same.p), same.t)) == k0
