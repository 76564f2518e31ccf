#!/usr/bin/env python3
"""Generate ``geometry_golden.json`` by importing the reference geometry module.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python tests/golden/make_geometry_golden.py

The reference module ``geometry_unified.py`` is NumPy-only and importable as checked in
(SURVEY.md §8c).  Only *data* (inputs and the values the reference returns) is stored; no reference
source text is written anywhere.  ``solver_fem.py`` itself is not importable (missing ``geometry``
/ ``config`` modules and scikit-fem), so the shift formula ``solver_fem.py:187-193`` is evaluated
here on the reference geometry's own attributes and stored as data as well.
"""
import json
import os
import sys

import numpy as np

REF = os.environ.get("PLFEM_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
import geometry_unified as gu  # noqa: E402  (reference module)

HERE = os.path.dirname(os.path.abspath(__file__))


def sigma_from(geom):
    # arithmetic of solver_fem.py:187-193 on the reference geometry's attributes
    n_core, n_clad = geom.n_core, geom.n_clad
    NA = np.sqrt(max(n_core ** 2 - n_clad ** 2, 1e-6))
    V = geom.k0 * np.mean(geom.core_radii) * NA
    b = max((1.0 - 2.405 / max(V, 2.41)) ** 2, 0.05)
    n_est = np.sqrt(n_clad ** 2 + b * (n_core ** 2 - n_clad ** 2))
    return float((geom.k0 * float(np.clip(n_est, n_clad + 0.05, n_core - 0.005))) ** 2)


def cplx(a):
    a = np.asarray(a)
    return {"re": a.real.tolist(), "im": a.imag.tolist()}


def main():
    out = {"layouts": [], "c1": {}, "pl_explicit": {}, "eps_probe": {}}
    rng = np.random.default_rng(20261004)

    # all 12 layouts (+ the 1+5 variant) at the self-test parameters geometry_unified.py:725
    cases = [(n, None) for n in gu.MCFGeometry.SUPPORTED_N] + [(6, "pentagon_center")]
    for n, variant in cases:
        for (pitch, r, n_core, lam) in [(8.0, 1.2, 1.53, 1.55), (8.0, 1.5, 1.535, 1.55), (6.0, 1.5, 1.535, 1.49)]:
            g = gu.MCFGeometry(n, pitch, r, n_core, 1.0, wavelength_um=lam, variant=variant)
            R = g.domain_radius
            px = rng.uniform(-R, R, 64)
            py = rng.uniform(-R, R, 64)
            # add points exactly on / next to core rims and in the PML
            cx, cy = g.positions[-1]
            px = np.concatenate([px, [cx, cx + r, cx + r * (1 + 1e-9), cx - r, 0.0, R - 0.1, R - 5.0]])
            py = np.concatenate([py, [cy, cy, cy, cy, 0.0, 0.0, 0.0]])
            ok, msg = g.validate()
            out["layouts"].append({
                "n_cores": n, "variant": variant, "pitch_um": pitch, "core_radius_um": r,
                "n_core": n_core, "n_clad": 1.0, "wavelength_um": lam,
                "config_type": g.config_type, "has_central_core": bool(g.has_central_core),
                "n_peripheral": int(g.n_peripheral), "R_ring": float(g.R_ring),
                "positions": g.positions.tolist(), "core_radii": g.core_radii.tolist(),
                "k0": float(g.k0), "V_number": float(g.V_number), "pitch": float(g.pitch),
                "cladding_radius": float(g.cladding_radius), "domain_radius": float(g.domain_radius),
                "packing_efficiency": float(g.packing_efficiency), "hash": g.hash,
                "valid": bool(ok), "valid_msg": msg, "sigma": sigma_from(g),
                "probe_x": px.tolist(), "probe_y": py.tolist(), "eps": cplx(g.epsilon(px, py)),
            })

    # C1 north-star geometry and its wavelength ladder (SURVEY.md §8c iii)
    for lam in (1.49, 1.55, 1.60, 1.65):
        g = gu.MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=lam)
        out["c1"][f"{lam:.2f}"] = {"k0": float(g.k0), "V_number": float(g.V_number),
                                   "sigma": sigma_from(g), "hash": g.hash,
                                   "domain_radius": float(g.domain_radius),
                                   "cladding_radius": float(g.cladding_radius)}
    g = gu.MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    xs = np.array([0, 1.5, 1.5000001, 8, 9.4, 21.9, 22.1, 31.9])
    out["eps_probe"] = {"x": xs.tolist(), "eps": cplx(g.epsilon(xs, np.zeros_like(xs)))}
    g19 = gu.MCFGeometry(19, 8.0, 1.5, 1.535, 1.0)
    out["c5_domain_radius"] = float(g19.domain_radius)

    # explicit-positions subclass, geometry_unified.py:637-678
    pos = gu.mcf_positions(7, 8.0)[0] + np.array([0.25, -0.125])
    radii = np.array([1.5, 1.4, 1.6, 1.5, 1.45, 1.55, 1.5])
    pl = gu.PhotonicLanternGeometry(7, "custom_7", pos, radii, 1.535, n_clad=1.0, wavelength=1.55)
    px = rng.uniform(-12, 12, 200)
    py = rng.uniform(-12, 12, 200)
    out["pl_explicit"] = {
        "positions": pos.tolist(), "core_radii": radii.tolist(), "n_core": 1.535, "wavelength": 1.55,
        "k0": float(pl.k0), "V_number": float(pl.V_number), "r_core": float(pl.r_core),
        "pitch": float(pl.pitch), "domain_radius": float(pl.domain_radius),
        "cladding_radius": float(pl.cladding_radius), "hash": pl.hash, "sigma": sigma_from(pl),
        "probe_x": px.tolist(), "probe_y": py.tolist(), "eps": cplx(pl.epsilon(px, py)),
    }

    # self-check constants the reference prints in its __main__ block (geometry_unified.py:766-772)
    g = gu.MCFGeometry(7, 8.0, 1.2, 1.53, 1.0)
    out["selfcheck"] = {"eps00": float(np.real(g.epsilon(np.array([0.0]), np.array([0.0])))[0]),
                        "eps100": float(np.real(g.epsilon(np.array([100.0]), np.array([0.0])))[0])}

    path = os.path.join(HERE, "geometry_golden.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
