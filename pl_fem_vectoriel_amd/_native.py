"""ctypes binding of ``libplfem_hip.so`` (C-ABI declared in ``include/plfem.h``).

There is no CPU fallback: if the shared library is missing, or no HIP device is visible when a
device context is requested, a ``RuntimeError`` is raised (the reference raises
``RuntimeError("scikit-fem requis")`` when its backend is missing, ``solver_fem.py:115-116``).
PyTorch is used only as the provider of device buffers and of the HIP stream.
"""
from __future__ import annotations

import ctypes
import time
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libplfem_hip.so")
HOOKS_PATH = os.path.join(_HERE, "libplfem_testhooks.so")

PLFEM_OK = 0
PLFEM_EINVAL, PLFEM_EMESH, PLFEM_EHIP, PLFEM_ENOCONV, PLFEM_ESTATE, PLFEM_ESINGULAR, PLFEM_ERESIDUAL = -1, -2, -3, -4, -5, -6, -7
BLOCKS = ("Axx", "Axy", "Ayx", "Ayy", "Minv", "Dxx", "Dxy", "Dyy")
INFO_NAMES = ("nv", "ne", "nedges", "N", "nsolve", "nnz", "levels", "nfronts", "front_doubles", "max_front",
              "solve_entries", "factor_flops", "t_numbering_us", "t_pattern_us", "t_tree_us", "t_fronts_us", "dofs_per_node",
              "arena_doubles")
POST_NAMES = ("norm", "div_energy", "core_x", "core_y", "all_x", "all_y")

# every symbol include/plfem.h declares (tests check the library exports all of them)
EXPORTS = (
    "plfem_symbolic_create", "plfem_symbolic_destroy", "plfem_symbolic_info", "plfem_symbolic_array_bytes",
    "plfem_symbolic_get", "plfem_workspace_bytes", "plfem_create", "plfem_destroy", "plfem_last_error", "plfem_synchronize",
    "plfem_assemble_hfield", "plfem_block_values_dev", "plfem_block_values_host", "plfem_spmv", "plfem_factor",
    "plfem_solve", "plfem_lanczos_shift_invert", "plfem_postprocess", "plfem_timings",
    "plfem_profile_begin", "plfem_profile_end",
    "plfem_mesh_edge_count", "plfem_mesh_refine",
    "plfem_residuals", "plfem_set_option", "plfem_symbolic_create_ex", "plfem_assemble_scalar", "plfem_cmt_coupling",
    "plfem_solve_modes", "plfem_modes_dev",
)
SOLVE_STATS = ("nconv", "n_opinv", "restarts", "max_rel_res", "n_block_solves", "true_residual_first", "true_residual", "refined",
               "pivot_perturbations", "assemble_us", "factor_us", "lanczos_us", "post_us", "upload_us", "residual_us", "call_us")
# the test hooks (include/plfem.h under PLFEM_TEST_HOOKS): exported by the add-on libplfem_testhooks.so ONLY
TEST_HOOK_EXPORTS = ("plfem_debug_factor_until", "plfem_debug_copy", "plfem_debug_solve_block", "plfem_debug_symeig",
                     "plfem_debug_symeig_band", "plfem_debug_set_perturb")
MAX_NCV = 320                       # PLFEM_MAX_NCV of include/plfem.h
PROF_SLOTS = ("k_fwd", "fwd_sweep", "bwd_sweep", "spmv_b")

_ARRAY_DTYPES = {
    "edof": np.int32, "tsorted": np.int32, "edges": np.int32, "doflocs": np.float64, "bmask": np.uint8,
    "interior": np.int32, "int_index": np.int32, "rowptr": np.int32, "colind": np.int32, "slot_row": np.int32,
    "nptr": np.int32, "nadj": np.int32, "nloc": np.uint8, "leaf_of_elem": np.int32, "leaf_elem_ptr": np.int32, "leaf_elems": np.int32, "epos": np.int32, "epos_leaf": np.int32,
    "owner": np.int32, "fs": np.int32, "fb": np.int32, "fs_true": np.int32, "fb_true": np.int32,
    "fnode_ptr": np.int64, "fnodes": np.int32, "cinv0": np.int32, "cinv1": np.int32, "foff": np.int64, "soff": np.int64, "prow": np.int32, "npos": np.int32,
}


from scipy.sparse.linalg import ArpackNoConvergence as _ArpackNoConvergence


class ArpackLikeNoConvergence(_ArpackNoConvergence, RuntimeError):
    """Raised when the Lanczos iteration does not converge.  The reference lets SciPy's
    ``ArpackNoConvergence`` propagate out of ``eigsh`` (``solver_fem.py:197``), so this IS one: a caller's
    ``except ArpackNoConvergence`` catches it, and ``.eigenvalues`` / ``.eigenvectors`` hold the current Ritz
    pairs as SciPy's do (``.eigenvectors`` here is the device tensor, row c = vector c)."""

    def __init__(self, msg, eigenvalues=None, eigenvectors=None):
        _ArpackNoConvergence.__init__(self, msg, eigenvalues, eigenvectors)


_lib = None
_hooks = None


def _tune_host_allocator(_force: bool = False) -> None:
    """Keep the host analysis out of the kernel's memory-map lock.

    ``plfem_symbolic_create`` allocates and frees ~20 MB of index arrays per cross-section (blocks of 0.1-3 MB) from
    several threads at once.  With glibc's defaults each of those blocks is its own ``mmap`` / ``munmap`` plus a page
    fault per 4 KB on first touch, all serialised on the process's memory-map lock: measured on the MI355X host, the
    analysis of C1 takes 5.5 ms that way and 4.3 ms when freed blocks stay in the heap.  So blocks below 32 MB are taken
    from the heap (``M_MMAP_THRESHOLD``) and the heap top is only returned to the system beyond 512 MB
    (``M_TRIM_THRESHOLD``).  This is process-wide policy and therefore OPT-IN: a drop-in library does not re-tune its
    host's allocator on its own.  An application that wants the faster analysis sets ``PLFEM_MALLOC_TUNE=1`` before the
    library is first loaded (``bench.py`` does) or calls this function / the two ``mallopt`` calls itself."""
    if os.environ.get("PLFEM_MALLOC_TUNE", "0") != "1" and not _force:
        return
    try:
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 32 << 20)       # M_MMAP_THRESHOLD (glibc's maximum)
        libc.mallopt(-1, 512 << 20)      # M_TRIM_THRESHOLD
    except (OSError, AttributeError):    # not glibc: nothing to tune
        pass


def load_library() -> ctypes.CDLL:
    """Load ``libplfem_hip.so``; raise loudly if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libplfem_hip.so requis: {LIB_PATH} not found — build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the eigenmode path.")
    # torch first: the wheel bundles its own libamdhip64 and the context's device memory comes from torch.  If this
    # library were loaded before torch, the process would hold TWO HIP runtimes (the system one bound here, torch's
    # own) and plfem_create would see no device in the one it is bound to.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    _tune_host_allocator()
    c_void_pp = ctypes.POINTER(ctypes.c_void_p)
    lib.plfem_symbolic_create.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_int32, ctypes.c_int32, c_void_pp, ctypes.c_char_p, ctypes.c_int32]
    lib.plfem_symbolic_create_ex.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32,
                                             ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_void_pp, ctypes.c_char_p, ctypes.c_int32]
    lib.plfem_symbolic_destroy.argtypes = [ctypes.c_void_p]
    lib.plfem_symbolic_destroy.restype = None
    lib.plfem_symbolic_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.plfem_symbolic_array_bytes.argtypes = [ctypes.c_void_p, ctypes.c_char_p]
    lib.plfem_symbolic_array_bytes.restype = ctypes.c_int64
    lib.plfem_symbolic_get.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p, ctypes.c_int64]
    lib.plfem_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64)]
    lib.plfem_create.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p,
                                 ctypes.c_int64, c_void_pp, ctypes.c_char_p, ctypes.c_int32]
    lib.plfem_destroy.argtypes = [ctypes.c_void_p]
    lib.plfem_destroy.restype = None
    lib.plfem_last_error.argtypes = [ctypes.c_void_p]
    lib.plfem_last_error.restype = ctypes.c_char_p
    lib.plfem_synchronize.argtypes = [ctypes.c_void_p]
    lib.plfem_assemble_hfield.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_double,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_double]
    lib.plfem_assemble_scalar.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_double,
                                          ctypes.c_double, ctypes.c_double]
    lib.plfem_cmt_coupling.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int32, ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
    lib.plfem_block_values_dev.argtypes = [ctypes.c_void_p, ctypes.c_int32, c_void_pp]
    lib.plfem_block_values_host.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    lib.plfem_spmv.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    lib.plfem_factor.argtypes = [ctypes.c_void_p, ctypes.c_double]
    lib.plfem_solve.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
    lib.plfem_lanczos_shift_invert.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_double,
                                               ctypes.c_int32, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p]
    lib.plfem_postprocess.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.plfem_timings.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.plfem_mesh_edge_count.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.POINTER(ctypes.c_int32), ctypes.c_char_p, ctypes.c_int32]
    lib.plfem_mesh_refine.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int32]
    lib.plfem_profile_begin.argtypes = [ctypes.c_void_p, ctypes.c_int32]
    lib.plfem_profile_end.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.plfem_residuals.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.plfem_set_option.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_double]
    lib.plfem_solve_modes.argtypes = ([ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32] + [ctypes.c_double] * 5 +
                                      [ctypes.c_int32, ctypes.c_int32, ctypes.c_double, ctypes.c_int32, ctypes.c_double,
                                       ctypes.c_double] + [ctypes.c_void_p] * 6)
    lib.plfem_modes_dev.argtypes = [ctypes.c_void_p, c_void_pp, ctypes.POINTER(ctypes.c_int32)]
    _lib = lib
    return lib


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def mesh_refine(p, t):
    """Uniform red refinement on the host side of the library (``plfem_mesh_refine``):
    returns ``(p2 (2, nv + nedges) float64, t2 (3, 4 ne) int32)``."""
    lib = load_library()
    p = np.ascontiguousarray(np.asarray(p, dtype=np.float64))
    t = np.ascontiguousarray(np.asarray(t, dtype=np.int32))
    if p.ndim != 2 or p.shape[0] != 2 or t.ndim != 2 or t.shape[0] != 3:
        raise ValueError("mesh.p must be (2, nv) and mesh.t (3, ne)")
    err = ctypes.create_string_buffer(512)
    nedges = ctypes.c_int32(0)
    rc = lib.plfem_mesh_edge_count(p.shape[1], t.shape[1], _ptr(p), _ptr(t), ctypes.byref(nedges), err, 512)
    if rc != PLFEM_OK:
        raise ValueError(f"plfem_mesh_edge_count failed ({rc}): {err.value.decode()}")
    p2 = np.empty((2, p.shape[1] + nedges.value), dtype=np.float64)
    t2 = np.empty((3, 4 * t.shape[1]), dtype=np.int32)
    rc = lib.plfem_mesh_refine(p.shape[1], t.shape[1], _ptr(p), _ptr(t), _ptr(p2), _ptr(t2), err, 512)
    if rc != PLFEM_OK:
        raise ValueError(f"plfem_mesh_refine failed ({rc}): {err.value.decode()}")
    return p2, t2


def load_test_hooks() -> ctypes.CDLL:
    """Load the add-on ``libplfem_testhooks.so`` (``plfem_debug_*``; tests and scripts only -- the product library
    exports none of them).  It links against ``libplfem_hip.so``, which is loaded first, so its entry points run the
    product library's own kernels on contexts the product library created."""
    global _hooks
    if _hooks is not None:
        return _hooks
    load_library()
    if not os.path.exists(HOOKS_PATH):
        raise RuntimeError(f"{HOOKS_PATH} not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'`")
    h = ctypes.CDLL(HOOKS_PATH)
    h.plfem_debug_factor_until.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
    h.plfem_debug_copy.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
    h.plfem_debug_solve_block.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32]
    h.plfem_debug_set_perturb.argtypes = [ctypes.c_void_p, ctypes.c_double]
    h.plfem_debug_symeig.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    h.plfem_debug_symeig_band.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                          ctypes.c_void_p, ctypes.c_void_p]
    _hooks = h
    return h


def debug_symeig(a, last_rows: int = -1):
    """Host eigensolver of the Lanczos drivers (test hook ``plfem_debug_symeig``): returns ``(w, V)`` with eigenvector i
    in row i of V — all n components, or only the last ``last_rows`` ones."""
    lib = load_test_hooks()
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    n = a.shape[0]
    if a.shape != (n, n):
        raise ValueError("square matrix expected")
    w = np.empty(n, dtype=np.float64)
    v = np.empty((n, n if last_rows < 0 else last_rows), dtype=np.float64)
    rc = lib.plfem_debug_symeig(n, _ptr(a), int(last_rows), _ptr(w), _ptr(v))
    if rc != PLFEM_OK:
        raise ValueError(f"plfem_debug_symeig failed ({rc})")
    return w, v


def debug_symeig_band(a, b: int, nsel: int):
    """Band path of the host eigensolver (test hook ``plfem_debug_symeig_band``): ``(w, V)`` with w ascending and the
    eigenvector of ``w[i]`` in row i of V for the ``nsel`` eigenvalues of largest magnitude (zero rows elsewhere)."""
    lib = load_test_hooks()
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    n = a.shape[0]
    if a.shape != (n, n):
        raise ValueError("square matrix expected")
    w = np.empty(n, dtype=np.float64)
    v = np.empty((n, n), dtype=np.float64)
    rc = lib.plfem_debug_symeig_band(n, int(b), _ptr(a), int(nsel), _ptr(w), _ptr(v))
    if rc != PLFEM_OK:
        raise ValueError(f"plfem_debug_symeig_band failed ({rc})")
    return w, v


def default_host_threads() -> int:
    """Workers of the symbolic analysis.  Measured on the MI355X host (256 logical CPUs, C1, round 4: teams, persistent
    pools): 3.0 ms with 12 workers, 3.2 with 16, 3.0 with 20, 2.55 with 24, 2.6 with 32; inside the cold-solve loop of
    ``bench.py`` 20.3 / 20.6 / 20.2 / 19.95 ms per step with 12 / 16 / 20 / 24 -- so 24 at most (round 3: 16), and no more
    than this process's share of the physical cores when several ranks run on the node (``LOCAL_WORLD_SIZE``, set by
    torchrun and by ``bench.py --gpus N``): the ranks of a sweep analyse their meshes at the same time.  (The workers spin
    for the ~3 ms of an analysis only: 24 of them stay far below the 16-CPU quota of a GPU box averaged over a step.)"""
    ncpu = os.cpu_count() or 1
    if ncpu < 64:
        return max(1, min(ncpu, 8))
    try:
        local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local_world = 1
    return max(2, min(24, (ncpu // 2) // local_world))


class Symbolic:
    """Mesh-only analysis (P2 numbering, CSR pattern, front tree).  Host only — needs no GPU."""

    def __init__(self, p, t, leaf_elems: int = 0, nthreads: int = 0, dofs_per_node: int = 2, dirichlet: bool = True):
        lib = load_library()
        p = np.ascontiguousarray(np.asarray(p, dtype=np.float64))
        t = np.ascontiguousarray(np.asarray(t, dtype=np.int32))
        if p.ndim != 2 or p.shape[0] != 2 or t.ndim != 2 or t.shape[0] != 3:
            raise ValueError("mesh.p must be (2, nv) and mesh.t (3, ne)")
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        if nthreads <= 0:
            nthreads = int(os.environ.get("PLFEM_HOST_THREADS", default_host_threads()))
        if leaf_elems <= 0:
            leaf_elems = int(os.environ.get("PLFEM_LEAF_ELEMS", 0))
        rc = lib.plfem_symbolic_create_ex(p.shape[1], t.shape[1], _ptr(p), _ptr(t), int(leaf_elems), int(nthreads),
                                          int(dofs_per_node), 1 if dirichlet else 0, ctypes.byref(h), err, 512)
        if rc != PLFEM_OK:
            raise ValueError(f"plfem_symbolic_create failed ({rc}): {err.value.decode()}")
        self._h = h
        self._lib = lib
        info = np.zeros(32, dtype=np.int64)
        lib.plfem_symbolic_info(h, _ptr(info))
        self.info = {k: int(info[i]) for i, k in enumerate(INFO_NAMES)}
        for k in ("nv", "ne", "nedges", "N", "nsolve", "nnz"):
            setattr(self, k, self.info[k])
        self.dofs_per_node = self.info["dofs_per_node"]

    def array(self, name: str) -> np.ndarray:
        nb = self._lib.plfem_symbolic_array_bytes(self._h, name.encode())
        if nb < 0:
            raise KeyError(name)
        out = np.empty(nb // np.dtype(_ARRAY_DTYPES[name]).itemsize, dtype=_ARRAY_DTYPES[name])
        rc = self._lib.plfem_symbolic_get(self._h, name.encode(), _ptr(out), ctypes.c_int64(nb))
        if rc != PLFEM_OK:
            raise RuntimeError(f"plfem_symbolic_get({name}) failed: {rc}")
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.plfem_symbolic_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """Device context bound to one :class:`Symbolic` (owns every device workspace)."""

    def __init__(self, sym: Symbolic, device: Optional[int] = None, max_ncv: int = 65, use_torch_stream: bool = True):
        import torch

        lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the eigenmode path has no CPU fallback")
        if device is None:
            device = torch.cuda.current_device()
        self.torch = torch
        self.device = int(device)
        self.tdev = torch.device("cuda", self.device)
        self.sym = sym          # keep the host analysis alive as long as the context
        self._lib = lib
        stream = torch.cuda.current_stream(self.tdev).cuda_stream if use_torch_stream else 0
        h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        # every device buffer of the context lives in ONE torch tensor: torch's caching allocator
        # recycles it when contexts come and go (cold solves in a loop pay no hipMalloc / hipFree)
        need = ctypes.c_int64(0)
        rc = lib.plfem_workspace_bytes(sym._h, int(max_ncv), ctypes.byref(need))
        if rc != PLFEM_OK:
            raise ValueError(f"plfem_workspace_bytes failed ({rc})")
        t0 = time.perf_counter()
        self.workspace = torch.empty(int(need.value) + 256, dtype=torch.uint8, device=self.tdev)
        self.t_workspace = time.perf_counter() - t0      # ~0 when the caching allocator recycles a block, ms for a hipMalloc
        base = self.workspace.data_ptr()
        aligned = (base + 255) & ~255
        rc = lib.plfem_create(sym._h, self.device, ctypes.c_void_p(stream), int(max_ncv), ctypes.c_void_p(aligned),
                              ctypes.c_int64(int(need.value)), ctypes.byref(h), err, 512)
        if rc in (PLFEM_EINVAL, PLFEM_EMESH):
            raise ValueError(f"plfem_create: {err.value.decode()}")
        if rc != PLFEM_OK:
            raise RuntimeError(f"plfem_create failed ({rc}): {err.value.decode()}")
        self._h = h
        self.max_ncv = int(max_ncv)
        self.N = sym.N
        self.dpn = sym.dofs_per_node
        self.n2 = self.dpn * sym.N            # length of every global vector

    # -- helpers ----------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc == PLFEM_OK:
            return
        msg = self._lib.plfem_last_error(self._h).decode()
        if rc in (PLFEM_EINVAL, PLFEM_EMESH):
            raise ValueError(f"{what}: {msg}")
        raise RuntimeError(f"{what} failed ({rc}): {msg}")

    def _cores(self, cores):
        c = np.ascontiguousarray(np.asarray(cores, dtype=np.float64).reshape(-1, 3))
        return c, c.shape[0]

    def empty(self, *shape):
        return self.torch.empty(*shape, dtype=self.torch.float64, device=self.tdev)

    # -- C-ABI calls ------------------------------------------------------------------------------
    def assemble(self, cores, eps_core, eps_clad, k0, alpha_p=1.0):
        c, n = self._cores(cores)
        self._check(self._lib.plfem_assemble_hfield(self._h, _ptr(c), n, float(eps_core), float(eps_clad),
                                                    float(k0), float(alpha_p)), "plfem_assemble_hfield")

    def assemble_scalar(self, cores, eps_core, eps_clad, k0):
        c, n = self._cores(cores)
        self._check(self._lib.plfem_assemble_scalar(self._h, _ptr(c), n, float(eps_core), float(eps_clad), float(k0)),
                    "plfem_assemble_scalar")

    def cmt_coupling(self, fields_i, fields_j, cores, eps_core, eps_clad):
        """``plfem_cmt_coupling``: raw[i, j] = E_i^T M_deps F_j, the squared norms of both field sets, mean(eps)."""
        n = fields_i.shape[0]
        c, nc = self._cores(cores)
        raw = np.zeros((n, n), dtype=np.float64)
        pi = np.zeros(n, dtype=np.float64)
        pj = np.zeros(n, dtype=np.float64)
        mean = ctypes.c_double(0.0)
        self._check(self._lib.plfem_cmt_coupling(self._h, int(n), ctypes.c_void_p(fields_i.data_ptr()),
                                                 ctypes.c_void_p(fields_j.data_ptr()), _ptr(c), nc, float(eps_core),
                                                 float(eps_clad), _ptr(raw), _ptr(pi), _ptr(pj), ctypes.byref(mean)),
                    "plfem_cmt_coupling")
        return raw.T.copy(), pi, pj, float(mean.value)      # C side is column major: raw_host[i + j n]

    def block_values(self, name: str) -> np.ndarray:
        out = np.empty(self.sym.nnz, dtype=np.float64)
        self._check(self._lib.plfem_block_values_host(self._h, BLOCKS.index(name), _ptr(out)), "plfem_block_values_host")
        return out

    def spmv(self, which: str, x):
        y = self.empty(self.n2)
        self._check(self._lib.plfem_spmv(self._h, 0 if which == "A" else 1, ctypes.c_void_p(x.data_ptr()),
                                         ctypes.c_void_p(y.data_ptr())), "plfem_spmv")
        return y

    def factor(self, sigma: float):
        self._check(self._lib.plfem_factor(self._h, float(sigma)), "plfem_factor")

    def solve(self, rhs, refine_steps: int = 0):
        x = self.empty(self.n2)
        self._check(self._lib.plfem_solve(self._h, ctypes.c_void_p(rhs.data_ptr()), ctypes.c_void_p(x.data_ptr()),
                                          int(refine_steps)), "plfem_solve")
        return x

    def lanczos(self, k, ncv, tol, maxiter, sigma):
        evals = np.empty(k, dtype=np.float64)
        evecs = self.empty(k, self.n2)
        stats = np.zeros(8, dtype=np.float64)
        rc = self._lib.plfem_lanczos_shift_invert(self._h, int(k), int(ncv), float(tol), int(maxiter), float(sigma),
                                                  _ptr(evals), ctypes.c_void_p(evecs.data_ptr()), _ptr(stats))
        st = {"nconv": int(stats[0]), "n_opinv": int(stats[1]), "restarts": int(stats[2]), "max_rel_res": float(stats[3]),
              "n_block_solves": int(stats[4])}       # 0: single-vector recurrence was used
        if rc == PLFEM_ENOCONV:
            raise ArpackLikeNoConvergence(self._lib.plfem_last_error(self._h).decode(), evals, evecs)
        self._check(rc, "plfem_lanczos_shift_invert")
        return evals, evecs, st

    def solve_modes(self, cores, eps_core, eps_clad, k0, alpha_p, sigma, k, ncv, tol, maxiter, residual_tol, tol_refined,
                    modes_host=None):
        """``plfem_solve_modes``: assembly, factorisation, eigen-solve, post-processing, a-posteriori check (with its
        refined second pass) and the copy of the interior mode vectors into ``modes_host`` (a pinned torch tensor of
        shape (k, dofs_per_node * nsolve), or None) in ONE call.  Returns ``(evals, post, frac_core, resid, stats)``."""
        c, n = self._cores(cores)
        evals = np.empty(k, dtype=np.float64)
        post = np.zeros((k, len(POST_NAMES)), dtype=np.float64)
        resid = np.zeros(k, dtype=np.float64)
        stats = np.zeros(len(SOLVE_STATS), dtype=np.float64)
        frac = ctypes.c_double(0.0)
        if modes_host is not None and (tuple(modes_host.shape) != (k, self.dpn * self.sym.nsolve) or not modes_host.is_contiguous()
                                       or modes_host.dtype != self.torch.float64 or modes_host.device.type != "cpu"):
            raise ValueError("modes_host must be a contiguous float64 CPU tensor of shape (k, dofs_per_node * nsolve)")
        rc = self._lib.plfem_solve_modes(self._h, _ptr(c), n, float(eps_core), float(eps_clad), float(k0), float(alpha_p),
                                         float(sigma), int(k), int(ncv), float(tol), int(maxiter), float(residual_tol),
                                         float(tol_refined), _ptr(evals), _ptr(post), ctypes.cast(ctypes.byref(frac), ctypes.c_void_p),
                                         _ptr(resid), ctypes.c_void_p(modes_host.data_ptr()) if modes_host is not None else None,
                                         _ptr(stats))
        st = {name: (float(stats[i]) if name.endswith("_us") or "res" in name else int(stats[i])) for i, name in enumerate(SOLVE_STATS)}
        st["refined"] = bool(st["refined"])
        if rc == PLFEM_ENOCONV:
            raise ArpackLikeNoConvergence(self._lib.plfem_last_error(self._h).decode(), evals, self.modes_dev())
        self._check(rc, "plfem_solve_modes")
        return evals, post, float(frac.value), resid, st

    def modes_dev(self):
        """The full-length vectors of the last eigen-solve as a torch tensor (a COPY of the context's own buffer, which
        the next factorisation reuses)."""
        ptr = ctypes.c_void_p()
        k = ctypes.c_int32(0)
        self._check(self._lib.plfem_modes_dev(self._h, ctypes.byref(ptr), ctypes.byref(k)), "plfem_modes_dev")

        class _View:
            __cuda_array_interface__ = {"shape": (int(k.value), int(self.n2)), "typestr": "<f8", "data": (int(ptr.value), True),
                                        "version": 2, "strides": None}
        return self.torch.as_tensor(_View(), device=self.tdev).clone()

    def postprocess(self, evecs, cores, want_interior: bool = True):
        k = evecs.shape[0]
        c, n = self._cores(cores)
        out = np.zeros((k, len(POST_NAMES)), dtype=np.float64)
        frac = ctypes.c_double(0.0)
        modes_int = self.empty(k, self.dpn * self.sym.nsolve) if want_interior else None
        self._check(self._lib.plfem_postprocess(self._h, int(k), ctypes.c_void_p(evecs.data_ptr()), _ptr(c), n,
                                                _ptr(out), ctypes.byref(frac),
                                                ctypes.c_void_p(modes_int.data_ptr()) if want_interior else None),
                    "plfem_postprocess")
        return out, float(frac.value), modes_int

    def timings(self):
        t = np.zeros(8, dtype=np.float64)
        self._check(self._lib.plfem_timings(self._h, _ptr(t)), "plfem_timings")
        return {"assemble_us": t[0], "factor_us": t[1], "lanczos_us": t[2], "post_us": t[3], "upload_us": t[4],
                "pivot_perturbations": int(t[5])}

    # -- test hooks (libplfem_testhooks.so; not in the product library) ---------------------------------------------
    def debug_factor_until(self, sigma, level, step, stage):
        self._check(load_test_hooks().plfem_debug_factor_until(self._h, float(sigma), int(level), int(step), int(stage)),
                    "plfem_debug_factor_until")

    def debug_copy(self, name: str, offset: int, count: int) -> np.ndarray:
        out = np.empty(int(count), dtype=np.float64)
        self._check(load_test_hooks().plfem_debug_copy(self._h, name.encode(), ctypes.c_int64(int(offset)),
                                                       ctypes.c_int64(int(count)), _ptr(out)), "plfem_debug_copy")
        return out

    def debug_solve_block(self, reps: int = 1, front_filter: int = 0):
        self._check(load_test_hooks().plfem_debug_solve_block(self._h, int(reps), int(front_filter)), "plfem_debug_solve_block")

    def debug_set_perturb(self, value: float):
        """Fault injection: D^-1 of the root front scaled by ``1 + value`` after every factorisation (0 = off)."""
        self._check(load_test_hooks().plfem_debug_set_perturb(self._h, float(value)), "plfem_debug_set_perturb")

    def profile_begin(self, max_launches: int = 4096):
        self._check(self._lib.plfem_profile_begin(self._h, int(max_launches)), "plfem_profile_begin")

    def profile_end(self):
        """Per slot of ``PROF_SLOTS``: ranges timed, their total HIP-event time and their algorithmic bytes.
        The ``k_fwd`` slot is also returned flat (``launches`` / ``total_us`` / ``bytes``)."""
        out = np.zeros((len(PROF_SLOTS), 3), dtype=np.float64)
        self._check(self._lib.plfem_profile_end(self._h, _ptr(out)), "plfem_profile_end")
        res = {"launches": int(out[0, 0]), "total_us": float(out[0, 1]), "bytes": float(out[0, 2])}
        res["slots"] = {name: {"ranges": int(out[i, 0]), "total_us": float(out[i, 1]), "bytes": float(out[i, 2])}
                        for i, name in enumerate(PROF_SLOTS)}
        return res

    def residuals(self, evals, evecs) -> np.ndarray:
        """``||A v_i - lambda_i B v_i|| / ||A v_i||`` against the assembled pencil (``plfem_residuals``)."""
        lam = np.ascontiguousarray(np.asarray(evals, dtype=np.float64))
        out = np.zeros(len(lam), dtype=np.float64)
        self._check(self._lib.plfem_residuals(self._h, len(lam), _ptr(lam), ctypes.c_void_p(evecs.data_ptr()), _ptr(out)),
                    "plfem_residuals")
        return out

    def set_option(self, name: str, value: float):
        self._check(self._lib.plfem_set_option(self._h, name.encode(), float(value)), "plfem_set_option")

    def synchronize(self):
        self._check(self._lib.plfem_synchronize(self._h), "plfem_synchronize")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.plfem_destroy(self._h)      # synchronises the stream before the workspace goes back to torch
            self._h = None
            self.workspace = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
