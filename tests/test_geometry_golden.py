"""Host geometry vs values produced by importing the reference geometry module
(tests/golden/make_geometry_golden.py -> geometry_golden.json)."""
import json
import os

import numpy as np
import pytest

from pl_fem_vectoriel_amd import MCFGeometry, PhotonicLanternGeometry, mcf_positions
from pl_fem_vectoriel_amd.solver_fem import shift_estimate

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "geometry_golden.json")))


def _c(d):
    return np.array(d["re"]) + 1j * np.array(d["im"])


@pytest.mark.parametrize("case", G["layouts"], ids=lambda c: f"{c['config_type']}-{c['pitch_um']}-{c['wavelength_um']}")
def test_layout_matches_reference(case):
    g = MCFGeometry(case["n_cores"], case["pitch_um"], case["core_radius_um"], case["n_core"], case["n_clad"],
                    wavelength_um=case["wavelength_um"], variant=case["variant"])
    assert g.config_type == case["config_type"]
    assert g.has_central_core == case["has_central_core"] and g.n_peripheral == case["n_peripheral"]
    np.testing.assert_array_equal(g.positions, np.array(case["positions"]))      # bit-exact positions
    np.testing.assert_array_equal(g.core_radii, np.array(case["core_radii"]))
    for k in ("k0", "V_number", "pitch", "cladding_radius", "domain_radius", "packing_efficiency", "R_ring"):
        assert getattr(g, k) == case[k], k
    assert g.hash == case["hash"]
    assert g.validate() == (case["valid"], case["valid_msg"])
    assert shift_estimate(g) == case["sigma"]
    eps = g.epsilon(np.array(case["probe_x"]), np.array(case["probe_y"]))
    np.testing.assert_array_equal(eps, _c(case["eps"]))


def test_c1_ladder_and_probe():
    for lam, ref in G["c1"].items():
        g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=float(lam))
        assert g.k0 == ref["k0"] and g.V_number == ref["V_number"] and g.hash == ref["hash"]
        assert shift_estimate(g) == ref["sigma"]
        assert g.domain_radius == ref["domain_radius"] == 32.0 and g.cladding_radius == 20.0
    g = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0)
    x = np.array(G["eps_probe"]["x"])
    np.testing.assert_array_equal(g.epsilon(x, np.zeros_like(x)), _c(G["eps_probe"]["eps"]))
    assert MCFGeometry(19, 8.0, 1.5, 1.535, 1.0).domain_radius == G["c5_domain_radius"] == 43.8
    # survey anchors (SURVEY.md §8c iii)
    assert g.k0 == 4.053667940115862 and shift_estimate(g) == 26.150716554571733


def test_explicit_positions_form():
    ref = G["pl_explicit"]
    pl = PhotonicLanternGeometry(7, "custom_7", np.array(ref["positions"]), np.array(ref["core_radii"]),
                                 ref["n_core"], n_clad=1.0, wavelength=ref["wavelength"])
    for k in ("k0", "V_number", "r_core", "pitch", "domain_radius", "cladding_radius"):
        assert getattr(pl, k) == ref[k], k
    assert pl.hash == ref["hash"] and shift_estimate(pl) == ref["sigma"]
    np.testing.assert_array_equal(pl.epsilon(np.array(ref["probe_x"]), np.array(ref["probe_y"])), _c(ref["eps"]))


def test_readme_keyword_form_and_errors():
    pl = PhotonicLanternGeometry(arrangement="hexagonal_1plus6_7", core_radius_um=1.5, pitch_um=8.0, n_core=1.535,
                                 n_clad=1.0, wavelength_nm=1550)
    ref = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    np.testing.assert_allclose(pl.positions, ref.positions)
    assert abs(pl.k0 - ref.k0) < 1e-15 and pl.n_cores == 7
    with pytest.raises(ValueError):
        mcf_positions(10, 8.0)                                   # geometry_unified.py:188
    with pytest.raises(ValueError):
        MCFGeometry(7, 8.0, 1.5, 1.0, 1.0)                       # delta_n too small, geometry_unified.py:235
    with pytest.raises(ValueError):
        PhotonicLanternGeometry(arrangement="nope", core_radius_um=1.5, pitch_um=8.0, n_core=1.5)
    g = MCFGeometry(7, 8.0, 1.2, 1.53, 1.0)                      # self-check of geometry_unified.py:766-772
    assert np.real(g.epsilon(np.array([0.0]), np.array([0.0])))[0] == G["selfcheck"]["eps00"]
    assert np.real(g.epsilon(np.array([100.0]), np.array([0.0])))[0] == G["selfcheck"]["eps100"]
