"""Triangle meshes for the eigenmode path (host side, outside the timed hot path).

The solver only needs ``mesh.p`` (2, nv) float64 and ``mesh.t`` (3, ne) int; the
reference hands it a scikit-fem ``MeshTri`` built by ``MeshGenerator._generate_mesh``
(reference ``mesh.py:222-340``).  This module provides

* :class:`TriMesh`          — the duck-type (``p``, ``t``, ``refined()``), columns of ``t`` sorted
  ascending as ``MeshTri`` does on construction;
* :func:`lantern_point_cloud` — the reference point recipe ``mesh.py:232-297`` (Cartesian grid +
  per-core polar rings + PML ring, radius filter, round(8), unique);
* :func:`generate_mesh`     — recipe -> ``scipy.spatial.Delaunay('QJ Pp')`` (``mesh.py:303``) ->
  drop zero-area triangles -> ``levels`` uniform red refinements.

Deliberate deviations from the reference, both documented in DESIGN.md:

* zero-area triangles produced by ``QJ`` on collinear hull points are dropped
  (``|det J| < 1e-10``); the reference would feed them to the P2 assembly and obtain Inf/NaN;
* the number of uniform refinements is an explicit argument, because the thresholds the reference
  reads (``config.mesh_min_points`` / ``mesh_target_points``, ``mesh.py:313-314``) live in a
  ``SimulationConfig`` that is absent from the reference checkout.
"""
from __future__ import annotations

import numpy as np

_LOCAL_EDGES = ((0, 1), (1, 2), (0, 2))


class TriMesh:
    """``p`` (2, nv) float64, ``t`` (3, ne) int32 with every column sorted ascending."""

    def __init__(self, p, t):
        self.p = np.ascontiguousarray(np.asarray(p, dtype=np.float64))
        t = np.sort(np.asarray(t, dtype=np.int64), axis=0)
        if self.p.ndim != 2 or self.p.shape[0] != 2 or t.ndim != 2 or t.shape[0] != 3:
            raise ValueError("mesh.p must be (2, nv) and mesh.t (3, ne)")
        if t.size and (t.min() < 0 or t.max() >= self.p.shape[1]):
            raise ValueError("mesh.t refers to a vertex outside mesh.p")
        self.t = np.ascontiguousarray(t.astype(np.int32))

    @property
    def nvertices(self) -> int:
        return self.p.shape[1]

    @property
    def nelements(self) -> int:
        return self.t.shape[1]

    def edges(self):
        """Unique sorted vertex pairs (2, nedges) in lexicographic order and ``t2f`` (3, ne)."""
        t = self.t.astype(np.int64)
        pairs = np.hstack([t[[i, j]] for (i, j) in _LOCAL_EDGES])
        nv = self.p.shape[1]
        key = pairs[0] * np.int64(nv) + pairs[1]
        ukey, inv = np.unique(key, return_inverse=True)
        return np.vstack([ukey // nv, ukey % nv]), inv.reshape(3, t.shape[1])

    def refined(self, times: int = 1) -> "TriMesh":
        """Uniform red refinement; new vertex id = nv + edge id (as ``MeshTri.refined()``, ``mesh.py:321``)."""
        m = self
        for _ in range(int(times)):
            f, t2f = m.edges()
            nv = m.p.shape[1]
            newp = np.hstack([m.p, 0.5 * (m.p[:, f[0]] + m.p[:, f[1]])])
            t = m.t.astype(np.int64)
            e = t2f + nv
            newt = np.hstack([np.vstack([t[0], e[0], e[2]]), np.vstack([t[1], e[0], e[1]]),
                              np.vstack([t[2], e[2], e[1]]), np.vstack([e[0], e[1], e[2]])])
            m = TriMesh(newp, newt)
        return m


def lantern_point_cloud(geometry, refinement: float = 1.0) -> np.ndarray:
    """Point recipe of ``MeshGenerator._generate_mesh`` (reference ``mesh.py:232-297``) -> (2, npts)."""
    R = float(geometry.domain_radius)
    n_base = max(int(25 + 20 * refinement), 16)
    g = np.linspace(-R, R, n_base, dtype=np.float64)
    X, Y = np.meshgrid(g, g)
    chunks = [np.vstack([X.ravel(), Y.ravel()])]

    theta = np.linspace(0, 2 * np.pi, max(int(16 * refinement), 12), endpoint=False)
    positions = np.atleast_2d(np.asarray(getattr(geometry, "positions",
                                                 getattr(geometry, "core_positions", np.zeros((1, 2))))))
    radii = np.asarray(geometry.core_radii)
    n_int = max(int(14 * refinement), 10)
    n_itf = max(int(18 * refinement), 14)
    for (cx, cy), r in zip(positions, radii):
        for rr in (np.linspace(0, r * 0.95, n_int), np.linspace(r * 0.90, r * 1.20, n_itf)):
            Rg, Tg = np.meshgrid(rr, theta)
            chunks.append(np.vstack([cx + Rg.ravel() * np.cos(Tg.ravel()),
                                     cy + Rg.ravel() * np.sin(Tg.ravel())]))
    pml_start = R - geometry.pml_thickness * 1.1
    if pml_start > 0:
        th = np.linspace(0, 2 * np.pi, max(int(36 * refinement), 24), endpoint=False)
        rr = np.linspace(pml_start, R * 0.98, max(int(18 * refinement), 12))
        Rg, Tg = np.meshgrid(rr, th)
        chunks.append(np.vstack([Rg.ravel() * np.cos(Tg.ravel()), Rg.ravel() * np.sin(Tg.ravel())]))
    pts = np.hstack(chunks)
    pts = pts[:, np.linalg.norm(pts, axis=0) <= R * 1.01]
    pts = np.round(pts.T, decimals=8).T
    return np.unique(pts, axis=1)


def generate_mesh(geometry, refinement: float = 1.0, levels: int = 1, drop_tol: float = 1e-10) -> TriMesh:
    """Synthetic lantern mesh: recipe -> Delaunay('QJ Pp') -> drop |det J| < tol -> red-refine."""
    from scipy.spatial import Delaunay

    pts = lantern_point_cloud(geometry, refinement)
    tri = Delaunay(pts.T, qhull_options="QJ Pp")
    p = tri.points.T
    t = tri.simplices.T
    p0, p1, p2 = p[:, t[0]], p[:, t[1]], p[:, t[2]]
    det = (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p2[0] - p0[0]) * (p1[1] - p0[1])
    t = t[:, np.abs(det) >= drop_tol]
    return TriMesh(p, t).refined(levels)


def unit_square_mesh(n: int = 4) -> TriMesh:
    """Structured test mesh of [0,1]^2 with 2 n^2 triangles (used by unit tests / smoke)."""
    g = np.linspace(0.0, 1.0, n + 1)
    X, Y = np.meshgrid(g, g)
    p = np.vstack([X.ravel(), Y.ravel()])
    idx = np.arange((n + 1) * (n + 1)).reshape(n + 1, n + 1)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel()
    t = np.hstack([np.vstack([a, b, d]), np.vstack([a, d, c])])
    return TriMesh(p, t)
