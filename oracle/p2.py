"""ORACLE (test infrastructure — never imported by the product path).

CPU restatement of the scikit-fem behaviour the reference relies on at ``solver_fem.py:126``
(``Basis(mesh, ElementTriP2())``) and ``mesh.py:308,321`` (``MeshTri``, ``refined()``).

scikit-fem is a third-party dependency of the reference that is absent from ``/root/reference``
and from this image (declared only as ``scikit-fem>=6.0`` in the reference ``README.md:122-129``,
no pin).  Its published algorithm is restated here from first principles:

* ``MeshTri`` sorts every element's vertex ids ascending on construction (``sort_t=True``);
* facets (edges) are numbered by the lexicographic rank of their sorted vertex pair taken over the
  local edges (0,1), (1,2), (0,2) (``np.unique(..., axis=1)``);
* ``ElementTriP2`` DOFs: vertex DOFs ``0..nv-1`` = vertex ids, then one DOF per edge ``nv + edge``;
  ``element_dofs`` rows 0-2 = vertices, rows 3-5 = edges (0,1), (1,2), (0,2);
* reference basis  phi0 = 1-3x-3y+2x^2+4xy+2y^2, phi1 = 2x^2-x, phi2 = 2y^2-y,
  phi3 = 4x-4x^2-4xy, phi4 = 4xy, phi5 = 4y-4xy-4y^2;
* default integration order ``2*maxdeg = 4`` -> the 6-point degree-4 D3-symmetric positive rule
  (Dunavant / Strang-Fix), which is unique as a point set;
* affine map ``x = p0 + J xi`` with ``J = [p1-p0, p2-p0]``, ``grad = J^-T grad_hat``,
  ``dx = |det J| w``;
* ``basis.get_dofs().all()`` = all vertex + edge DOFs on facets owned by exactly one element.

PARITY PINNING: the reference ships no tests or golden vectors for this path, and
scikit-fem cannot be run here, so the numbering / quadrature restatement is pinned by closed-form
known-answer tests only (tests/test_oracle_p2.py): parity of the *assembly* half is "unpinned"
against the real scikit-fem; the geometry half is pinned by values produced by importing the
reference (tests/golden/geometry_golden.json); the eigensolver half is SciPy itself.
"""
from __future__ import annotations

import numpy as np

# ----------------------------------------------------------------------------------------------
# 6-point degree-4 rule on the reference triangle (0,0),(1,0),(0,1); weights sum to 1/2
# ----------------------------------------------------------------------------------------------
_A = 0.445948490915965
_B = 0.091576213509771
_WA = 0.223381589678011 / 2.0
_WB = 0.109951743655322 / 2.0
QUAD_X = np.array([[_A, _A], [1 - 2 * _A, _A], [_A, 1 - 2 * _A],
                   [_B, _B], [1 - 2 * _B, _B], [_B, 1 - 2 * _B]]).T.copy()   # (2, 6)
QUAD_W = np.array([_WA, _WA, _WA, _WB, _WB, _WB])

LOCAL_EDGES = ((0, 1), (1, 2), (0, 2))


def p2_basis(x, y):
    """phi (6, ...) and reference gradients dphi (6, 2, ...) of ElementTriP2."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    one = np.ones_like(x)
    phi = np.stack([1 - 3 * x - 3 * y + 2 * x * x + 4 * x * y + 2 * y * y,
                    2 * x * x - x,
                    2 * y * y - y,
                    4 * x - 4 * x * x - 4 * x * y,
                    4 * x * y,
                    4 * y - 4 * x * y - 4 * y * y])
    dx = np.stack([-3 + 4 * x + 4 * y, 4 * x - 1, 0 * one, 4 - 8 * x - 4 * y, 4 * y, -4 * y])
    dy = np.stack([-3 + 4 * x + 4 * y, 0 * one, 4 * y - 1, -4 * x, 4 * x, 4 - 4 * x - 8 * y])
    return phi, np.stack([dx, dy], axis=1)


PHI_Q, DPHI_Q = p2_basis(QUAD_X[0], QUAD_X[1])      # (6, 6) [basis, qp], (6, 2, 6)


class MeshTriLite:
    """Minimal stand-in for ``skfem.MeshTri``: ``p`` (2, nv) float64, ``t`` (3, ne) sorted columns."""

    def __init__(self, p, t, sort_t: bool = True):
        self.p = np.ascontiguousarray(np.asarray(p, dtype=np.float64))
        t = np.asarray(t, dtype=np.int64)
        self.t = np.ascontiguousarray(np.sort(t, axis=0) if sort_t else t)
        self._edges = None

    # edges / facets ---------------------------------------------------------------------------
    def _build_edges(self):
        t = self.t
        pairs = np.hstack([t[[i, j]] for (i, j) in LOCAL_EDGES])            # (2, 3 ne), already sorted
        pairs = np.sort(pairs, axis=0)
        nv = self.p.shape[1]
        key = pairs[0] * np.int64(nv) + pairs[1]                             # lexicographic rank key
        ukey, inverse, counts = np.unique(key, return_inverse=True, return_counts=True)
        self._edges = (np.vstack([ukey // nv, ukey % nv]),
                       inverse.reshape(3, t.shape[1]), counts)

    @property
    def facets(self):
        if self._edges is None:
            self._build_edges()
        return self._edges[0]

    @property
    def t2f(self):
        if self._edges is None:
            self._build_edges()
        return self._edges[1]

    def boundary_facets(self):
        if self._edges is None:
            self._build_edges()
        return np.nonzero(self._edges[2] == 1)[0]

    # uniform red refinement (skfem MeshTri.refined(): new vertex = nv + facet id) --------------
    def refined(self, times: int = 1) -> "MeshTriLite":
        m = self
        for _ in range(int(times)):
            p, t, f, t2f = m.p, m.t, m.facets, m.t2f
            nv = p.shape[1]
            newp = np.hstack([p, 0.5 * (p[:, f[0]] + p[:, f[1]])])
            e = t2f + nv                                                    # mid-edge vertex ids
            newt = np.hstack([np.vstack([t[0], e[0], e[2]]),
                              np.vstack([t[1], e[0], e[1]]),
                              np.vstack([t[2], e[2], e[1]]),
                              np.vstack([e[0], e[1], e[2]])])
            m = MeshTriLite(newp, newt)
        return m


class _DofsView:
    def __init__(self, dofs):
        self._d = dofs

    def all(self):
        return self._d


class P2Basis:
    """Stand-in for ``skfem.Basis(mesh, ElementTriP2())`` (what ``solver_fem.py`` reads of it)."""

    def __init__(self, mesh: MeshTriLite):
        self.mesh = mesh
        nv = mesh.p.shape[1]
        f = mesh.facets
        self.nv = nv
        self.nedges = f.shape[1]
        self.N = nv + self.nedges
        self.element_dofs = np.vstack([mesh.t, nv + mesh.t2f])               # (6, ne)
        self.doflocs = np.hstack([mesh.p, 0.5 * (mesh.p[:, f[0]] + mesh.p[:, f[1]])])
        # affine maps
        p, t = mesh.p, mesh.t
        p0, p1, p2 = p[:, t[0]], p[:, t[1]], p[:, t[2]]
        self.J = np.stack([p1 - p0, p2 - p0], axis=1)                        # J[r, c, e] = d x_r / d xi_c
        self.detJ = self.J[0, 0] * self.J[1, 1] - self.J[0, 1] * self.J[1, 0]
        self.absdet = np.abs(self.detJ)
        inv = np.empty_like(self.J)
        with np.errstate(divide="ignore", invalid="ignore"):
            inv[0, 0] = self.J[1, 1] / self.detJ
            inv[0, 1] = -self.J[0, 1] / self.detJ
            inv[1, 0] = -self.J[1, 0] / self.detJ
            inv[1, 1] = self.J[0, 0] / self.detJ
        self.invJ = inv                                                       # (2, 2, e)
        self.dx = self.absdet[:, None] * QUAD_W[None, :]                      # (ne, 6)
        # global quadrature points w.x: (2, ne, 6)
        self.qx = p0[:, :, None] + np.einsum("rce,cq->req", self.J, QUAD_X)
        # basis values (6, ne, 6) and physical gradients (6, 2, ne, 6):  grad = J^-T grad_hat
        ne = t.shape[1]
        self.phi = np.broadcast_to(PHI_Q[:, None, :], (6, ne, 6))
        self.grad = np.einsum("cre,icq->ireq", inv, DPHI_Q)

    def get_dofs(self):
        bf = self.mesh.boundary_facets()
        f = self.mesh.facets
        d = np.unique(np.concatenate([f[0, bf], f[1, bf], self.nv + bf]))
        return _DofsView(d)
