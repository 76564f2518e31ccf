#!/usr/bin/env python3
"""Generates ``hfield_golden.npz``: the assembled pencil of ``TrueVectorialMaxwellSolver.assemble_hfield_system``
(reference ``solver_fem.py:122-169``) and the eigenvalues of ``solve_vectorial_modes`` (``:171-239``) on two small
meshes, as computed by the ORACLE (``oracle/hfield.py`` + ``oracle/p2.py``, SciPy ``eigsh`` with the reference's
arguments) in the build container -- SURVEY.md section 7 step 1.

What this fixture is and is not: the reference's solver cannot run (``solver_fem.py`` does not import as checked in and
its assembly backend scikit-fem is absent), so these are NOT reference outputs: parity of the
assembly half stays "unpinned" (DESIGN.md section 2).  The fixture freezes the oracle -- numbering, quadrature set,
eps sampling at the quadrature points of interface-cut elements, block signs, Dirichlet set, shift, request size -- so
that (a) a change of the oracle shows up as a diff against committed numbers instead of silently moving every parity
test with it, and (b) the HIP path is compared with committed data, not only with code that runs beside it.

Only data is stored: the meshes (p, t -- hand-made coarse meshes, the MeshGenerator recipe has no setting under 5 000
elements for 7 cores), the geometry parameters, CSR arrays of A, B, Dxx, Dxy, Dyy, M_inv, the shift, the 22 eigenvalues
and the n_eff list.

Run once here:  python3 tests/golden/make_hfield_golden.py
"""
import os
import sys

import numpy as np
from scipy.spatial import Delaunay

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hfield  # noqa: E402
from oracle.p2 import MeshTriLite  # noqa: E402
from pl_fem_vectoriel_amd import MCFGeometry  # noqa: E402
from pl_fem_vectoriel_amd.geometry import ARRANGEMENTS  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hfield_golden.npz")

CASES = {
    "hex7": dict(arrangement="hexagonal_1plus6_7", pitch_um=8.0, core_radius_um=1.5, n_core=1.535, n_clad=1.0,
                 wavelength_um=1.55, grid_um=3.2, n_modes=10),
    "lin2": dict(arrangement="linear_2", pitch_um=8.0, core_radius_um=1.5, n_core=1.535, n_clad=1.0,
                 wavelength_um=1.60, grid_um=4.0, n_modes=10),
}


def geometry_of(c):
    n, variant = ARRANGEMENTS[c["arrangement"]]
    return MCFGeometry(n, c["pitch_um"], c["core_radius_um"], c["n_core"], c["n_clad"], wavelength_um=c["wavelength_um"],
                       variant=variant)


def coarse_mesh(g, grid_um):
    """A few hundred points: a square grid clipped to the domain disc, two rings and the centre per core (the discs
    cut through elements: the quadrature points of one element see both materials), a ring on the outer boundary."""
    R = float(g.domain_radius)
    ax = np.arange(-R, R + 1e-9, grid_um)
    X, Y = np.meshgrid(ax, ax)
    pts = [np.stack([X.ravel(), Y.ravel()], 1)]
    pts[0] = pts[0][np.hypot(pts[0][:, 0], pts[0][:, 1]) < 0.93 * R]
    ang8 = np.arange(8) * (2 * np.pi / 8)
    for (cx, cy), r in zip(np.asarray(g.core_positions), np.asarray(g.core_radii)):
        pts.append(np.array([[cx, cy]]))
        pts.append(np.stack([cx + 0.55 * r * np.cos(ang8), cy + 0.55 * r * np.sin(ang8)], 1))
        pts.append(np.stack([cx + 1.25 * r * np.cos(ang8 + 0.3), cy + 1.25 * r * np.sin(ang8 + 0.3)], 1))
    nb = 40
    angb = np.arange(nb) * (2 * np.pi / nb)
    pts.append(np.stack([R * np.cos(angb), R * np.sin(angb)], 1))
    p = np.unique(np.round(np.vstack(pts), 8), axis=0)
    # drop grid points that crowd a core's rings (keeps the triangles around the interfaces well shaped)
    keep = np.ones(len(p), bool)
    for (cx, cy), r in zip(np.asarray(g.core_positions), np.asarray(g.core_radii)):
        d = np.hypot(p[:, 0] - cx, p[:, 1] - cy)
        on_ring = np.isclose(d, 0.55 * r, atol=1e-6) | np.isclose(d, 1.25 * r, atol=1e-6) | (d < 1e-9)
        keep &= on_ring | (d > 1.9 * r)
    p = p[keep]
    t = Delaunay(p).simplices
    area2 = np.abs((p[t[:, 1], 0] - p[t[:, 0], 0]) * (p[t[:, 2], 1] - p[t[:, 0], 1])
                   - (p[t[:, 2], 0] - p[t[:, 0], 0]) * (p[t[:, 1], 1] - p[t[:, 0], 1]))
    t = t[area2 > 1e-10]
    return np.ascontiguousarray(p.T), np.ascontiguousarray(t.T.astype(np.int32))


def main():
    out = {}
    for name, c in CASES.items():
        g = geometry_of(c)
        p, t = coarse_mesh(g, c["grid_um"])
        mesh = MeshTriLite(p, t)
        A, B, basis, Dxx, Dyy, Dxy, Minv = hfield.assemble_hfield_system(g, mesh)      # scikit-fem's loop shape, 9 forms
        A2, B2, *_ = hfield.assemble_hfield_system_fused(g, mesh)
        assert abs(A - A2).max() <= 1e-12 * abs(A).max() and abs(B - B2).max() <= 1e-13 * abs(B).max()
        modes, stats = hfield.solve_vectorial_modes(g, mesh, c["n_modes"], fused=False, return_raw=True)
        # which elements does a core interface cut (quadrature points in both materials)?
        em = hfield.element_matrices(g, basis)
        ratio = em["mass_eps_inv"].sum(axis=(1, 2)) / em["mass"].sum(axis=(1, 2))
        ne_cut = int(np.sum((ratio < 1.0 - 1e-9) & (ratio > 1.0 / c["n_core"] ** 2 + 1e-9)))
        print(f"{name}: nv {p.shape[1]} ne {t.shape[1]} N {basis.N} nnz(A) {A.nnz} interface-cut elements {ne_cut} "
              f"sigma {stats['sigma']:.12f} modes {len(modes)} n_eff {modes[0]['n_eff']:.9f} .. {modes[-1]['n_eff']:.9f}")
        out[f"{name}_params"] = np.array([c["pitch_um"], c["core_radius_um"], c["n_core"], c["n_clad"], c["wavelength_um"], c["n_modes"]])
        out[f"{name}_arrangement"] = np.array(c["arrangement"])
        out[f"{name}_p"], out[f"{name}_t"] = p, t
        for key, M in (("A", A), ("B", B), ("Dxx", Dxx), ("Dxy", Dxy), ("Dyy", Dyy), ("Minv", Minv)):
            M = M.tocsr()
            M.sort_indices()
            out[f"{name}_{key}_data"], out[f"{name}_{key}_indices"], out[f"{name}_{key}_indptr"] = M.data, M.indices.astype(np.int32), M.indptr.astype(np.int32)
        out[f"{name}_sigma"] = np.array(stats["sigma"])
        out[f"{name}_beta_sq"] = np.sort(np.asarray(stats["beta_sq"]))
        out[f"{name}_n_eff"] = np.array([m["n_eff"] for m in modes])
        out[f"{name}_boundary"] = np.asarray(basis.get_dofs().all(), dtype=np.int32)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
