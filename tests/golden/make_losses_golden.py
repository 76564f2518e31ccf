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
  EnhancedLossCalculator.calculate_sectional_losses (the scalar route of calculate_physical_losses, losses.py:828-865) and
      _calculate_pdl_realistic     losses.py:74-440, 470-541   (geometry and design parameters duck-typed: recorded inputs)
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
    # sectional model of the scalar route (losses.py:74-440, :470-541): calculate_sectional_losses is importable and
    # duck-types its design parameters (L_mux, coupling_uniformity, L_taper, n_taper, L_MMF, packing_efficiency,
    # pitch_ratio, wavelength) and its geometry (positions, n_core / n_clad): both are INPUTS recorded here
    sect = []
    hexa = [[0.0, 0.0]] + [[8.0 * np.cos(a), 8.0 * np.sin(a)] for a in np.arange(6) * np.pi / 3]
    geoms = {"hex7": {"positions": hexa, "n_core": 1.535, "n_clad": 1.0},
             "lin3": {"positions": [[-8.0, 0.0], [0.0, 0.0], [8.0, 0.0]], "n_core": 1.535, "n_clad": 1.0},
             "two": {"positions": [[-4.0, 0.0], [4.0, 0.0]], "n_core": 1.5, "n_clad": 1.44}}
    dps = {"default": {"L_mux": 200.0, "coupling_uniformity": 0.95, "L_taper": 375.0, "n_taper": 1.0, "L_MMF": 100.0,
                       "packing_efficiency": 0.12, "pitch_ratio": 2.6667, "wavelength": 1550.0},
           "dense_short": {"L_mux": 100.0, "coupling_uniformity": 0.8, "L_taper": 150.0, "n_taper": 0.3, "L_MMF": 0.5,
                           "packing_efficiency": 0.9, "pitch_ratio": 3.5, "wavelength": 1490.0},
           "mid": {"L_mux": 500.0, "coupling_uniformity": 1.0, "L_taper": 2000.0, "n_taper": 2.0, "L_MMF": 1e6,
                   "packing_efficiency": 0.6, "pitch_ratio": 5.0, "wavelength": 1650.0}}
    sect_specs = [("scalar_seed21_n8_hex7", scalar_modes(21, 8), "hex7", "default", "mux", 1550.0),
                  ("scalar_seed22_n18_hex7_demux", scalar_modes(22, 18), "hex7", "default", "demux", 1550.0),
                  ("scalar_seed23_n5_lin3_1490", scalar_modes(23, 5), "lin3", "dense_short", "mux", 1490.0),
                  ("scalar_seed24_n2_two_1650", scalar_modes(24, 2), "two", "mid", "demux", 1650.0),
                  ("scalar_seed25_n1", scalar_modes(25, 1), "hex7", "default", "mux", 1600.0),
                  ("vectorial_through_the_sectional_model", synthetic_modes(8, 9), "hex7", "default", "mux", 1550.0)]
    spread = scalar_modes(26, 6)
    for i, m in enumerate(spread):                      # no near-degenerate n_eff: the ptp branch of the birefringence term
        m["n_eff"] = 1.45 - 0.01 * i
        m["confinement"] = 0.005 if i == 5 else m["confinement"]
    sect_specs.append(("scalar_spread_n_eff_one_unconfined", spread, "lin3", "mid", "mux", 1550.0))
    for name, modes, gname, dname, direction, wl in sect_specs:
        res = ref.EnhancedLossCalculator.calculate_sectional_losses(as_arrays(modes), SimpleNamespace(**geoms[gname]),
                                                                    SimpleNamespace(**dps[dname]), direction, wl)
        sect.append({"name": name, "modes": modes, "geometry": geoms[gname], "design_params": dps[dname], "direction": direction,
                     "wavelength_nm": wl, "sectional_losses": res,
                     "pdl_realistic": ref.EnhancedLossCalculator._calculate_pdl_realistic(as_arrays(modes), SimpleNamespace(**geoms[gname]), wl)})
    sect_errors = {"no_modes": ref.EnhancedLossCalculator.calculate_sectional_losses([], None, None)}
    complex_beta = [{"confinement": 0.9, "beta": [4.8, 2e-7]}, {"confinement": 0.6, "beta": [4.7, 0.0]}]
    rad_c = ref.EnhancedLossCalculator._calculate_radiation_loss(
        [{"confinement": m["confinement"], "beta": complex(*m["beta"])} for m in complex_beta], 1600.0)
    doc = {"generator": "tests/golden/make_losses_golden.py (imports /root/reference/losses.py)",
           "cases": cases, "errors": errors, "scalar_cases": scal, "sectional_cases": sect, "sectional_errors": sect_errors,
           "radiation_complex_beta": {"modes": complex_beta, "wavelength_nm": 1600.0, "value": rad_c}}
    with open(OUT, "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    print(f"wrote {OUT}: {len(cases)} vectorial cases, {len(scal)} scalar cases, {len(sect)} sectional cases")


if __name__ == "__main__":
    main()
