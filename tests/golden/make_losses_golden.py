#!/usr/bin/env python3
"""Generates ``losses_golden.json``: inputs and outputs of the reference's vectorial loss models, produced by
IMPORTING the reference (``/root/reference/losses.py`` is NumPy-only and importable) in the build
container.  Only data is stored: the mode records fed in (the synthetic recipe of the reference's own self-test,
``losses.py:1233-1250``, for several seeds and sizes, plus edge cases) and what the reference returned for them.

Run once here:  python3 tests/golden/make_losses_golden.py
(The GPU box has no /root/reference; the committed JSON travels instead.)

What is called, as it stands in the reference:
  VectorialLossCalculator.calculate_vectorial_losses(modes, geometry, design_params, direction, wavelength_nm)
      losses.py:1012-1104   (design_params is duck-typed there: only .d_polymer, .L_taper, .n_taper are read,
                             losses.py:1114,1152-1153 -- a SimpleNamespace INPUT, recorded in the fixture)
  EnhancedLossCalculator._calculate_crosstalk_vectorial / _calculate_crosstalk / _calculate_pdl_vectorial /
      _calculate_radiation_loss / _calculate_crosstalk_scalar     losses.py:445-467, 546-720
``LossCalculator.calculate_physical_losses`` itself cannot run: it imports ``config.PhotonicLanternDesignParameters``,
which the checkout does not contain (losses.py:760); its vectorial route is the composition of the
functions above plus the demux asymmetry of losses.py:786-811, which the port restates and tests/ checks by
composition."""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
import losses as ref  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "losses_golden.json")


def synthetic_modes(seed, n, ne0=1.20, dne=0.003):
    """The recipe of the reference's self-test (losses.py:1233-1250) with seed / count / spacing as parameters."""
    rng = np.random.default_rng(seed)
    modes = []
    for k in range(n):
        Px = float(rng.uniform(0.3, 0.7))
        Py = 1.0 - Px
        modes.append({
            "n_eff": float(ne0 - k * dne + rng.normal(0, 1e-4)),
            "beta": float((2 * np.pi / 1.55) * (ne0 - k * dne)),
            "P_x": Px, "P_y": Py,
            "PDL_dB": float(10 * np.log10(max(Px, Py) / min(Px, Py))),
            "polarization": "Hybrid",
            "confinement": float(rng.uniform(0.55, 0.72)),
            "core_overlap": 0.60, "div_ratio": 0.02, "is_vectorial": True, "method": "H-field_V18.10",
        })
    return modes


def scalar_modes(seed, n, size=40):
    rng = np.random.default_rng(seed)
    modes = []
    for k in range(n):
        modes.append({"n_eff": float(1.45 - 0.002 * k + (1e-5 if k == 2 else 0.0) * rng.normal()),
                      "beta": float(4.05 * (1.45 - 0.002 * k)), "field_vector": rng.standard_normal(size).tolist(),
                      "confinement": float(rng.uniform(0.5, 0.99)), "is_vectorial": False})
    return modes


def as_arrays(modes):
    out = []
    for m in modes:
        m = dict(m)
        if "field_vector" in m:
            m["field_vector"] = np.asarray(m["field_vector"])
        out.append(m)
    return out


def main():
    cases = []
    dp_default = {"d_polymer": 2.0, "L_taper": 375.0, "n_taper": 1.0}
    specs = [
        ("self_test_seed42_n7", synthetic_modes(42, 7), dp_default, "mux", 1550.0),
        ("self_test_seed42_n7_demux", synthetic_modes(42, 7), dp_default, "demux", 1550.0),
        ("seed1_n22", synthetic_modes(1, 22, 1.2622, 6.6e-5), dp_default, "mux", 1550.0),
        ("seed2_n22_1490", synthetic_modes(2, 22, 1.2622, 6.6e-5), dp_default, "demux", 1490.0),
        ("seed3_n3_long_taper", synthetic_modes(3, 3), {"d_polymer": 5.0, "L_taper": 2000.0, "n_taper": 0.3}, "mux", 1650.0),
        ("seed4_n2", synthetic_modes(4, 2), dp_default, "mux", 1600.0),
        ("seed5_n1", synthetic_modes(5, 1), dp_default, "mux", 1550.0),
    ]
    # edge cases: one polarisation empty, poorly confined modes, equal n_eff
    edge = synthetic_modes(6, 5)
    for m in edge:
        m["P_x"], m["P_y"], m["PDL_dB"] = 1e-31, 1.0, 50.0
    specs.append(("edge_px_vanishing", edge, dp_default, "mux", 1550.0))
    edge2 = synthetic_modes(7, 6)
    for i, m in enumerate(edge2):
        m["confinement"] = 0.004 + 0.001 * i
        m["n_eff"] = 1.3
    specs.append(("edge_unconfined_degenerate", edge2, dp_default, "demux", 1550.0))
    for name, modes, dp, direction, wl in specs:
        geometry = SimpleNamespace()            # never read by calculate_vectorial_losses
        res = ref.VectorialLossCalculator.calculate_vectorial_losses(modes, geometry, SimpleNamespace(**dp), direction, wl)
        E = ref.EnhancedLossCalculator
        cases.append({
            "name": name, "modes": modes, "design_params": dp, "direction": direction, "wavelength_nm": wl,
            "vectorial_losses": res,
            "crosstalk_vectorial": E._calculate_crosstalk_vectorial(modes),
            "crosstalk": E._calculate_crosstalk(modes),
            "pdl_vectorial": E._calculate_pdl_vectorial(modes),
            "radiation_loss": E._calculate_radiation_loss(modes, wl),
        })
    # error paths of calculate_vectorial_losses (losses.py:1034-1039)
    errors = {
        "no_modes": ref.VectorialLossCalculator.calculate_vectorial_losses([], None, SimpleNamespace(**dp_default)),
        "not_vectorial": ref.VectorialLossCalculator.calculate_vectorial_losses(
            [{"is_vectorial": False}], None, SimpleNamespace(**dp_default)),
    }
    scal = []
    for seed, n in ((11, 4), (12, 2), (13, 1)):
        modes = scalar_modes(seed, n)
        scal.append({"modes": modes, "crosstalk_scalar": ref.EnhancedLossCalculator._calculate_crosstalk_scalar(as_arrays(modes)),
                     "crosstalk": ref.EnhancedLossCalculator._calculate_crosstalk(as_arrays(modes))})
    complex_beta = [{"confinement": 0.9, "beta": [4.8, 2e-7]}, {"confinement": 0.6, "beta": [4.7, 0.0]}]
    rad_c = ref.EnhancedLossCalculator._calculate_radiation_loss(
        [{"confinement": m["confinement"], "beta": complex(*m["beta"])} for m in complex_beta], 1600.0)
    doc = {"generator": "tests/golden/make_losses_golden.py (imports /root/reference/losses.py)",
           "cases": cases, "errors": errors, "scalar_cases": scal,
           "radiation_complex_beta": {"modes": complex_beta, "wavelength_nm": 1600.0, "value": rad_c}}
    with open(OUT, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    print(f"wrote {OUT}: {len(cases)} vectorial cases, {len(scal)} scalar cases")


if __name__ == "__main__":
    main()
