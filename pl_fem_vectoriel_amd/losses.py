"""Consumer of the eigenmode path: the reference's vectorial loss models, fed with the mode
records ``TrueVectorialMaxwellSolver.solve_vectorial_modes`` returns.

Reference: ``losses.py`` — ``LossCalculator.calculate_physical_losses`` (vectorial route ``:742-825``; scalar route
``:828-865`` -> ``EnhancedLossCalculator.calculate_sectional_losses`` ``:74-440`` with ``_calculate_pdl_realistic``
``:470-541``, the consumer of ``ScalarHelmholtzSolver``'s records, added in round 3),
``VectorialLossCalculator`` (``:996-1221``) and the helpers it reaches in ``EnhancedLossCalculator``
(``_calculate_pdl_vectorial :445-467``, ``_calculate_crosstalk_vectorial :546-619``, ``_calculate_crosstalk_scalar
:622-663``, ``_calculate_crosstalk :666-690``, ``_calculate_radiation_loss :693-720``).  These are O(#modes) closed-form
heuristics on the mode dict keys ``n_eff, beta, P_x, P_y, PDL_dB, confinement`` (the key contract of SURVEY.md row a9);
the reference keeps them in NumPy on the host and so does this port — there is nothing here for a GPU.

Pinned by ``tests/golden/losses_golden.json`` = outputs of the imported reference for the same inputs
(``tests/golden/make_losses_golden.py``).  One class the reference's route needs is absent from its checkout
(``config.PhotonicLanternDesignParameters``, imported at ``losses.py:760``): :class:`PhotonicLanternDesignParameters`
below carries the fields ``_build_design_params`` fills (``losses.py:956-989``).
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Dict, List, Sequence

import numpy as np

logger = logging.getLogger("pl_v18.losses")

_TINY = 1e-30
_DB_PER_NEPER_POWER = 8.685889638          # 20 / ln 10


@dataclass
class PhotonicLanternDesignParameters:
    """Field list of the reference's (absent) class as ``_build_design_params`` fills it (``losses.py:956-989``)."""
    N_cores: int
    has_central_core: bool
    config_type: str
    geometry_config: str
    n_peripheral_cores: int
    R_ring: float
    packing_efficiency: float
    pitch: float
    pitch_min: float
    pitch_ratio: float
    wavelength: float
    r_core_SM: float
    r_clad_SM: float
    n_core_SM: float
    n_clad_SM: float
    V_SM: float
    NA_SM: float
    MFD: float
    n_eff_LP01: float
    r_core_MM: float
    V_MM: float
    NA_MM: float
    M_max: int
    n_polymer: float
    d_polymer: float
    coupling_uniformity: float
    L_mux: float
    L_taper: float
    L_MMF: float
    L_total: float
    n_taper: float
    taper_profile: str


def _column(modes: Sequence[Dict], key: str, default=None) -> np.ndarray:
    if default is None:
        return np.array([m[key] for m in modes], dtype=float)
    return np.array([m.get(key, default) for m in modes], dtype=float)


def _ratio_db(a: float, b: float) -> float:
    """10 log10(max / (min + tiny)) of two powers."""
    hi, lo = (a, b) if a >= b else (b, a)
    return 10.0 * np.log10(hi / (lo + _TINY))


class EnhancedLossCalculator:
    """The helper estimators the vectorial route uses (``losses.py:445-720``)."""

    @staticmethod
    def _calculate_pdl_vectorial(modes: List[Dict]) -> float:
        px = float(np.sum(_column(modes, "P_x", 1.0)))
        py = float(np.sum(_column(modes, "P_y", 1.0)))
        if px < _TINY and py < _TINY:
            return 0.1
        return float(np.clip(_ratio_db(px, py), 0.0, 50.0))

    @staticmethod
    def _calculate_crosstalk_vectorial(modes: List[Dict]) -> float:
        """Spectral-spread proxy (``losses.py:546-619``): -10 - 20 Q - 5 CV - 5 Gamma, clipped to [-40, -15] dB."""
        if len(modes) < 2:
            return -25.0
        ne = np.sort(_column(modes, "n_eff"))
        conf = _column(modes, "confinement", 0.5)
        gaps = np.diff(ne)
        lo, hi = float(ne[0]), float(ne[-1])
        guide = max((hi + 0.01) - (lo - 0.002), 1e-6)             # estimated n_core - n_clad
        Q = float(np.clip((hi - lo) / guide, 0.0, 1.0))
        if gaps.size > 1:
            cv = float(np.std(gaps)) / (float(np.mean(gaps)) + 1e-12)
            cv_norm = float(np.clip(cv / 2.0, 0.0, 1.0))
        else:
            cv_norm = 0.5
        guided = conf > 0.01
        gamma = float(np.mean(conf[guided])) if np.any(guided) else 0.5
        return float(np.clip(-10.0 - 20.0 * Q - 5.0 * cv_norm - 5.0 * gamma, -40.0, -15.0))

    @staticmethod
    def _calculate_crosstalk_scalar(modes: List[Dict]) -> float:
        """Largest normalised field overlap between two modes (``losses.py:622-663``)."""
        if len(modes) < 2:
            return -70.0
        fields = [m.get("field_vector") for m in modes]
        powers = [float(np.real(np.vdot(f, f))) if f is not None else 0.0 for f in fields]
        worst = 0.0
        for i, fi in enumerate(fields):
            if fi is None or powers[i] < 1e-12:
                continue
            for j in range(i + 1, len(fields)):
                fj = fields[j]
                if fj is None or powers[j] < 1e-12:
                    continue
                worst = max(worst, float(np.abs(np.vdot(fi, fj)) ** 2 / (powers[i] * powers[j] + 1e-16)))
        if worst == 0.0:
            return -70.0
        xt = -10.0 * np.log10(worst + 1e-15)
        ne = np.sort(_column(modes, "n_eff"))
        if ne.size > 1:
            closest = float(np.min(np.diff(ne)))
            if closest < 1e-4:                                     # near-degenerate pair
                xt -= 15.0 + (1e-4 - closest) * 1e6
        return float(np.clip(xt, -70.0, -15.0))

    @staticmethod
    def _calculate_crosstalk(modes: List[Dict]) -> float:
        if not modes:
            return -70.0
        if modes[0].get("is_vectorial", False):
            return EnhancedLossCalculator._calculate_crosstalk_vectorial(modes)
        return EnhancedLossCalculator._calculate_crosstalk_scalar(modes)

    @staticmethod
    def _calculate_radiation_loss(modes: List[Dict], wavelength_nm: float) -> float:
        """dB/m: from Im(beta) when the propagation constant is complex, else a confinement penalty (``losses.py:693-720``)."""
        scale = 1550.0 / wavelength_nm
        per_mode = []
        for m in modes:
            conf, beta = m["confinement"], m["beta"]
            if np.iscomplexobj(beta) and abs(beta.imag) > 1e-9:
                per_mode.append(2.0 * abs(beta.imag) * 1e6 * _DB_PER_NEPER_POWER * scale)
                continue
            loss = max(0.0, 1.0 - conf) * 100.0
            if conf < 0.95:
                loss += (0.95 - conf) * 250.0
            per_mode.append(loss)
        return float(np.mean(per_mode)) if per_mode else 0.0


    # ---- sectional model (polymer -> taper -> MMF), reference losses.py:74-440 ------------------------------------------
    @staticmethod
    def _calculate_pdl_realistic(modes: List[Dict], geometry, wavelength_nm: float) -> float:
        """PDL estimate for scalar modes (no P_x / P_y): modal birefringence of near-degenerate n_eff, second-moment
        asymmetry of the core positions, a coupling term, the confinement spread, a wavelength factor (``:470-541``)."""
        if len(modes) < 2:
            return 0.3
        n_effs = np.array([float(m["n_eff"]) for m in modes])
        desc = np.sort(n_effs)[::-1]
        gaps = np.abs(desc[:-1] - desc[1:])
        gaps = gaps[gaps < 5e-4]
        if gaps.size:
            k0 = 2.0 * np.pi / (wavelength_nm * 1e-9)
            pdl_biref = 4.343 * k0 * np.mean(gaps) * 375e-6
        else:
            pdl_biref = np.ptp(n_effs) * 800.0
        pdl_geom = 0.0
        positions = getattr(geometry, "positions", None)
        if positions is not None and len(positions) >= 3:
            c = np.array(positions)
            c = c - np.mean(c, axis=0)
            ixx, iyy, ixy = np.sum(c[:, 0] ** 2), np.sum(c[:, 1] ** 2), np.sum(c[:, 0] * c[:, 1])
            disc = np.sqrt(((ixx - iyy) / 2.0) ** 2 + ixy ** 2)
            i_max, i_min = (ixx + iyy) / 2.0 + disc, (ixx + iyy) / 2.0 - disc
            pdl_geom = abs(i_max - i_min) / (i_max + i_min + 1e-12) * 4.0
        pdl_coupling = 0.15 * np.log10(len(modes) + 1)
        if wavelength_nm < 1530:
            wl = 1.0 + (1530.0 - wavelength_nm) / 1000.0
        elif wavelength_nm > 1565:
            wl = 1.0 + (wavelength_nm - 1565.0) / 1000.0
        else:
            wl = 1.0
        pdl_conf = np.std(np.array([m["confinement"] for m in modes])) * 2.0
        return float(np.clip((pdl_biref + pdl_geom + pdl_coupling + pdl_conf) * wl, 0.05, 6.0))

    @staticmethod
    def _calculate_polymer_section(modes: List[Dict], geometry, design_params, wavelength_nm: float) -> Dict:
        """``:190-246``: coupling mismatch + confinement + 0.5 dB/m over L_mux; MDL from the confinement spread."""
        confs = np.array([m["confinement"] for m in modes])
        avg = float(np.mean(confs[confs > 0.01])) if np.any(confs > 0.01) else 0.5
        il = 0.5 * (1.0 - design_params.coupling_uniformity) - 10.0 * np.log10(max(avg, 1e-6)) + 0.5 * (design_params.L_mux * 1e-6)
        mdl = (-10.0 * np.log10(max(np.min(confs), 1e-9) / (np.max(confs) + 1e-12)) + 3.0 * np.std(confs)) if len(confs) >= 2 else 0.0
        if modes[0].get("is_vectorial", False):
            pdl = EnhancedLossCalculator._calculate_pdl_vectorial(modes)
        else:
            pdl = EnhancedLossCalculator._calculate_pdl_realistic(modes, geometry, wavelength_nm)
        return {"IL": float(np.clip(il, 0.0, 10.0)), "MDL": float(np.clip(mdl, 0.0, 5.0)), "PDL": float(np.clip(pdl, 0.05, 3.0))}

    @staticmethod
    def _calculate_taper_section(modes: List[Dict], geometry, design_params, wavelength_nm: float) -> Dict:
        """``:252-321``: adiabaticity (L_beat = 150 um), 0.5 dB/m, residual radiation; MDL from the three best / worst
        confined modes; PDL from a 1e-5 birefringence over the taper."""
        L, n_taper = design_params.L_taper, design_params.n_taper
        eta = 1.0 - np.exp(-L / (150.0 * max(n_taper, 0.5)))
        confs = np.array([m["confinement"] for m in modes])
        conf_mean = float(np.mean(confs)) if len(confs) else 0.9
        il = -10.0 * np.log10(max(eta, 1e-6)) + 0.5 * (L * 1e-6) + (max(0.0, 1.0 - conf_mean) * 0.5 + 0.05 * np.log10(len(modes) + 1))
        if len(confs) >= 2:
            asc = np.sort(confs)
            mdl = float(np.clip(-10.0 * np.log10(np.mean(asc[:3]) / (np.mean(asc[-3:]) + 1e-12)), 0.0, 3.0))
        else:
            mdl = 0.0
        pdl = 4.343 * (2.0 * np.pi / (wavelength_nm * 1e-3)) * 1e-5 * L
        return {"IL": float(np.clip(il, 0.0, 8.0)), "MDL": float(np.clip(mdl, 0.0, 3.0)), "PDL": float(np.clip(pdl, 0.01, 2.0))}

    @staticmethod
    def _calculate_mmf_section(modes: List[Dict], geometry, design_params, wavelength_nm: float) -> Dict:
        """``:327-358``: 0.2 dB/km of silica + 0.3 dB splice; MDL = PDL = 0.05."""
        L = design_params.L_MMF
        if L < 1.0:
            return {"IL": 0.0, "MDL": 0.0, "PDL": 0.0}
        return {"IL": float(np.clip(0.2 * (L * 1e-9) + 0.3, 0.0, 5.0)), "MDL": 0.05, "PDL": 0.05}

    @staticmethod
    def _calculate_global_metrics(polymer: Dict, taper: Dict, mmf: Dict, modes: List[Dict], geometry, design_params) -> Dict:
        """``:364-439``: IL adds, MDL adds in quadrature, PDL adds; crosstalk, coupling degradation from the spread of the
        FEM modes, packing / pitch penalties, radiation loss, mean confinement."""
        il = polymer["IL"] + taper["IL"] + mmf["IL"]
        mdl = np.sqrt(polymer["MDL"] ** 2 + taper["MDL"] ** 2 + mmf["MDL"] ** 2)
        pdl = polymer["PDL"] + taper["PDL"] + mmf["PDL"]
        xt = EnhancedLossCalculator._calculate_crosstalk(modes)
        if len(modes) >= 2:
            confs = np.array([m["confinement"] for m in modes])
            n_effs = np.array([float(m["n_eff"]) for m in modes])
            n_core = getattr(geometry, "core_index", getattr(geometry, "n_core", 1.53))
            n_clad = getattr(geometry, "clad_index", getattr(geometry, "n_clad", 1.0))
            cv = float(np.std(confs) / (np.mean(confs) + 1e-9))
            spread = float(np.ptp(n_effs) / max(n_core - n_clad, 1e-6))
            degradation = float(np.clip(cv * 1.5 + spread * 0.8 + float(max(0.0, 0.70 - float(np.min(confs)))) * 2.0, 0.0, 5.0))
        else:
            degradation = 5.0
        packing, pitch_ratio = design_params.packing_efficiency, design_params.pitch_ratio
        pack_pen = (0.5 - packing) * 3.0 if packing < 0.5 else ((packing - 0.85) * 2.0 if packing > 0.85 else 0.0)
        geom_pen = pack_pen + abs(pitch_ratio - 3.5) * 0.2
        valid = [m["confinement"] for m in modes if m["confinement"] > 0]
        return {
            "IL_total": float(np.clip(il, 0.0, 40.0)), "MDL_total": float(np.clip(mdl, 0.0, 10.0)),
            "PDL_total": float(np.clip(pdl, 0.05, 10.0)), "Total_Loss": float(il),
            "Efficiency": float(np.clip(10.0 ** (-il / 10.0), 0.0, 1.0)), "Crosstalk": float(xt),
            "crosstalk_penalty": float(np.clip(max(0.0, -20.0 - xt) * 0.1, 0.0, 5.0)),
            "coupling_degradation": float(np.clip(degradation, 0.0, 5.0)), "geometry_penalty": float(np.clip(geom_pen, 0.0, 5.0)),
            "radiation_loss_dB_per_m": float(EnhancedLossCalculator._calculate_radiation_loss(modes, design_params.wavelength)),
            "avg_confinement": float(np.mean(valid)) if valid else 0.0,
        }

    @staticmethod
    def calculate_sectional_losses(modes: List[Dict], geometry, design_params, direction: str = "mux",
                                   wavelength_nm: float = 1550.0) -> Dict:
        """``:74-184``: losses per section and the cumulated metrics; errors come back as ``{'success': False, ...}``."""
        if not modes:
            return {"success": False, "error": "no modes"}
        try:
            E = EnhancedLossCalculator
            po = E._calculate_polymer_section(modes, geometry, design_params, wavelength_nm)
            ta = E._calculate_taper_section(modes, geometry, design_params, wavelength_nm)
            mm = E._calculate_mmf_section(modes, geometry, design_params, wavelength_nm)
            g = E._calculate_global_metrics(po, ta, mm, modes, geometry, design_params)
            return {
                "IL_polymer": po["IL"], "MDL_polymer": po["MDL"], "PDL_polymer": po["PDL"],
                "IL_taper": ta["IL"], "MDL_taper": ta["MDL"], "PDL_taper": ta["PDL"],
                "IL_MMF": mm["IL"], "MDL_MMF": mm["MDL"], "PDL_MMF": mm["PDL"],
                "IL_total": g["IL_total"], "MDL_total": g["MDL_total"], "PDL_total": g["PDL_total"],
                "Total_Loss": g["Total_Loss"], "Efficiency": g["Efficiency"], "Crosstalk": g["Crosstalk"],
                "crosstalk_penalty": g["crosstalk_penalty"], "coupling_degradation": g["coupling_degradation"],
                "geometry_penalty": g["geometry_penalty"], "radiation_loss_dB_per_m": g["radiation_loss_dB_per_m"],
                "avg_confinement": g["avg_confinement"], "n_modes_used": len(modes), "direction": direction,
                "wavelength_nm": float(wavelength_nm), "success": True,
            }
        except Exception as e:                                    # noqa: BLE001 - the reference reports, it does not raise (:182-184)
            logger.error(f"Erreur calcul pertes sectionnées: {e}")
            return {"error": str(e), "success": False}


class VectorialLossCalculator:
    """Sectional losses (polymer, taper, MMF) with the PDL taken from the FEM powers P_x / P_y (``losses.py:996-1221``)."""

    @staticmethod
    def calculate_vectorial_losses(modes_vectorial: List[Dict], geometry, design_params, direction: str = "mux",
                                   wavelength_nm: float = 1550.0) -> Dict:
        if not modes_vectorial:
            return {"success": False, "error": "no modes"}
        if not modes_vectorial[0].get("is_vectorial", False):
            logger.warning("Modes non-vectoriels passés à VectorialLossCalculator")
            return {"success": False, "error": "modes not vectorial"}
        try:
            sections = {
                "polymer": VectorialLossCalculator._polymer_vectorial(modes_vectorial, design_params, wavelength_nm),
                "taper": VectorialLossCalculator._taper_vectorial(modes_vectorial, design_params, wavelength_nm),
                "MMF": VectorialLossCalculator._mmf_vectorial(modes_vectorial, design_params),
            }
        except Exception as exc:                       # noqa: BLE001 - the reference reports, never raises (losses.py:1102)
            logger.error(f"Erreur VectorialLossCalculator: {exc}")
            return {"success": False, "error": str(exc)}
        out: Dict = {"success": True, "is_vectorial": True}
        for name, sec in sections.items():
            out[f"IL_{name}"] = sec["IL"]
            out[f"MDL_{name}"] = sec["MDL"]
            out[f"PDL_{name}"] = sec["PDL"]
            out[f"PDL_x_{name}"] = sec["PDL_x"]
            out[f"PDL_y_{name}"] = sec["PDL_y"]
        p, t, f = sections["polymer"], sections["taper"], sections["MMF"]
        out["IL_total"] = float(np.clip(p["IL"] + t["IL"] + f["IL"], 0.0, 40.0))
        out["MDL_total"] = float(np.clip(np.sqrt(p["MDL"] ** 2 + t["MDL"] ** 2 + f["MDL"] ** 2), 0.0, 10.0))
        out["PDL_total"] = float(np.clip(p["PDL"] + t["PDL"] + f["PDL"], 0.05, 10.0))     # sections add in dB
        out["n_modes_used"] = len(modes_vectorial)
        out["direction"] = direction
        out["wavelength_nm"] = float(wavelength_nm)
        return out

    @staticmethod
    def _polymer_vectorial(modes_v, design_params, wavelength_nm: float) -> Dict:
        il = 0.2 * (design_params.d_polymer * 1e-6)                  # 0.2 dB/m (IP-Dip) over d_polymer um
        confs = [m["confinement"] for m in modes_v]
        mdl = 10.0 * np.log10(max(confs) / (min(confs) + 1e-12)) if len(confs) > 1 else 0.0
        px = float(np.sum([m.get("P_x", 1.0) for m in modes_v]))
        py = float(np.sum([m.get("P_y", 1.0) for m in modes_v]))
        pdl = _ratio_db(px, py) if (px > _TINY and py > _TINY) else 0.1
        return {"IL": float(np.clip(il, 0.0, 1.0)), "MDL": float(np.clip(mdl, 0.0, 2.0)),
                "PDL": float(np.clip(pdl, 0.05, 1.0)), "PDL_x": px, "PDL_y": py}

    @staticmethod
    def _taper_vectorial(modes_v, design_params, wavelength_nm: float) -> Dict:
        length, n_taper = design_params.L_taper, design_params.n_taper
        beat = 150.0
        eta = 1.0 - np.exp(-length / (beat * max(n_taper, 0.5)))          # adiabatic transfer
        il = -10.0 * np.log10(max(eta, 1e-6)) + 0.5 * (length * 1e-6)
        confs = np.array([m["confinement"] for m in modes_v])
        il += max(0.0, 1.0 - float(np.mean(confs))) * 0.5 + 0.05 * np.log10(len(modes_v) + 1)
        px = [m.get("P_x", 1.0) for m in modes_v]
        py = [m.get("P_y", 1.0) for m in modes_v]
        mdl = 10.0 * np.log10(1.0 + (np.var(px) + np.var(py)) / 2.0) if len(px) > 1 else 0.0
        each = [m.get("PDL_dB", 0.0) for m in modes_v]
        weights = [a + b for a, b in zip(px, py)]
        pdl = float(np.average(each, weights=weights)) if sum(weights) > 1e-12 else float(np.mean(each))
        k0 = 2.0 * np.pi / (wavelength_nm * 1e-3)
        pdl += 4.343 * k0 * 1e-5 * length                                   # taper birefringence 1e-5
        return {"IL": float(np.clip(il, 0.0, 10.0)), "MDL": float(np.clip(mdl, 0.0, 5.0)),
                "PDL": float(np.clip(pdl, 0.01, 3.0)), "PDL_x": float(np.sum(px)), "PDL_y": float(np.sum(py))}

    @staticmethod
    def _mmf_vectorial(modes_v, design_params) -> Dict:
        return {"IL": 0.32, "MDL": 0.05, "PDL": 0.05,
                "PDL_x": float(np.mean([m.get("P_x", 1.0) for m in modes_v])),
                "PDL_y": float(np.mean([m.get("P_y", 1.0) for m in modes_v]))}


class LossCalculator(EnhancedLossCalculator):
    """``calculate_physical_losses`` of the reference: vectorial route (``losses.py:742-825``) for the records of
    ``TrueVectorialMaxwellSolver``, scalar route (``:828-865``, through ``calculate_sectional_losses``) for the records of
    ``ScalarHelmholtzSolver``."""

    @staticmethod
    def calculate_physical_losses(modes: List[Dict], geometry, direction: str = "mux", wavelength_nm: float = 1550.0) -> Dict:
        if not (modes and modes[0].get("is_vectorial", False)):
            # scalar route (``losses.py:828-865``): the records of ScalarHelmholtzSolver through the sectional model
            params = LossCalculator._build_design_params(modes, geometry, wavelength_nm)
            full = EnhancedLossCalculator.calculate_sectional_losses(modes, geometry, params, direction, wavelength_nm)
            if not full.get("success", False):
                return {"success": False, "error": full.get("error", "unknown")}
            pdl = full["PDL_total"] * (1.02 if direction == "demux" else 1.0)
            return {"IL_dB": full["IL_total"], "MDL_dB": full["MDL_total"], "PDL_dB": float(np.clip(pdl, 0.05, 10.0)),
                    "crosstalk_dB": full["Crosstalk"], "radiation_loss_dB_per_m": full["radiation_loss_dB_per_m"],
                    "avg_confinement": full["avg_confinement"], "n_modes_used": full["n_modes_used"], "direction": direction,
                    "wavelength_nm": float(wavelength_nm), "is_vectorial": False, "success": True}
        params = LossCalculator._build_design_params(modes, geometry, wavelength_nm)
        res = VectorialLossCalculator.calculate_vectorial_losses(modes, geometry, params, direction, wavelength_nm)
        if not res.get("success", False):
            return {"success": False, "error": res.get("error", "unknown")}
        pdl = res["PDL_total"]
        if direction == "demux":
            # demultiplexing excites the slow, weakly confined modes first: PDL_demux > PDL_mux by 2-12 % (losses.py:786-811)
            each = np.sort(np.array([m.get("PDL_dB", 0.0) for m in modes]))
            spread = max(float(np.mean(each[-4:])) - float(np.mean(each[:4])), 0.0) if each.size >= 4 else 0.3
            confs = np.array([m.get("confinement", 0.5) for m in modes])
            cv = float(np.std(confs) / (np.mean(confs) + 1e-9))
            pdl = pdl * (1.0 + float(np.clip(0.04 + 0.06 * cv + 0.02 * spread, 0.02, 0.12)))
        confs = [m.get("confinement", 0.0) for m in modes]
        return {
            "IL_dB": res["IL_total"], "MDL_dB": res["MDL_total"], "PDL_dB": float(np.clip(pdl, 0.05, 10.0)),
            "crosstalk_dB": EnhancedLossCalculator._calculate_crosstalk_vectorial(modes),
            "radiation_loss_dB_per_m": EnhancedLossCalculator._calculate_radiation_loss(modes, wavelength_nm),
            "avg_confinement": float(np.mean(confs)) if confs else 0.0,
            "n_modes_used": res["n_modes_used"], "direction": direction, "wavelength_nm": float(wavelength_nm),
            "is_vectorial": True, "success": True,
        }

    @staticmethod
    def _build_design_params(modes: List[Dict], geometry, wavelength_nm: float) -> PhotonicLanternDesignParameters:
        """Design parameters from the actual geometry (``losses.py:871-989``)."""
        def scalar(value):
            return float(np.asarray(value).flat[0])

        n_cores = int(getattr(geometry, "n_cores", 3))
        radii = getattr(geometry, "core_radii", None)
        r_core = scalar(radii) if radii is not None else float(getattr(geometry, "r_core", 1.2))
        n_core = scalar(getattr(geometry, "core_index", getattr(geometry, "n_core", 1.535)))
        n_clad = scalar(getattr(geometry, "clad_index", getattr(geometry, "n_clad", 1.0)))
        k0 = scalar(getattr(geometry, "k0", 2.0 * np.pi / (wavelength_nm / 1000.0)))
        NA = float(np.sqrt(max(n_core ** 2 - n_clad ** 2, 1e-6)))
        V = getattr(geometry, "V_number", None)
        V = scalar(V) if V is not None else float(k0 * r_core * NA)
        Vc = max(V, 0.5)
        MFD = float(2.0 * r_core * (0.65 + 1.619 / Vc ** 1.5 + 2.879 / Vc ** 6))           # Marcuse
        positions = getattr(geometry, "positions", getattr(geometry, "core_positions", None))
        positions = list(positions) if positions is not None else None
        if positions and len(positions) >= 2:
            pts = np.array(positions, dtype=float)
            d = np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=2)
            pitch = float(np.min(d[np.triu_indices(len(pts), 1)]))
            R_ring = float(np.max(np.linalg.norm(pts, axis=1)))
        else:
            pitch = R_ring = 8.0
        packing = float(np.clip(n_cores * np.pi * r_core ** 2 / (np.pi * max(R_ring + r_core, 1.0) ** 2), 0.01, 0.90))
        has_central = bool(positions) and bool(np.any(np.linalg.norm(np.array(positions, dtype=float), axis=1) < 0.5 * r_core))
        kind = "hexagonal" if n_cores in (7, 19) else "circular"
        taper = getattr(geometry, "taper_length", None)
        taper = scalar(taper) if taper is not None else 0.0
        L_taper, L_mux = (taper, max(taper * 0.5, 100.0)) if taper > 0.0 else (375.0, 200.0)
        L_MMF = 100.0
        return PhotonicLanternDesignParameters(
            N_cores=n_cores, has_central_core=has_central, config_type=kind, geometry_config=f"{n_cores}-{kind}",
            n_peripheral_cores=n_cores - (1 if has_central else 0), R_ring=R_ring, packing_efficiency=packing,
            pitch=pitch, pitch_min=pitch, pitch_ratio=float(pitch / (2.0 * r_core + 1e-9)), wavelength=float(wavelength_nm),
            r_core_SM=r_core, r_clad_SM=62.5, n_core_SM=n_core, n_clad_SM=n_clad, V_SM=float(V), NA_SM=NA, MFD=MFD,
            n_eff_LP01=float(modes[0]["n_eff"]) if modes else float(n_core - 0.01), r_core_MM=25.0,
            V_MM=float(np.sqrt(n_cores) * V), NA_MM=0.22, M_max=max(int(n_cores * V ** 2 / 4), 1), n_polymer=n_core,
            d_polymer=2.0, coupling_uniformity=0.95, L_mux=L_mux, L_taper=L_taper, L_MMF=L_MMF,
            L_total=L_mux + L_taper + L_MMF, n_taper=1.0, taper_profile="exponential")


__all__ = ["EnhancedLossCalculator", "VectorialLossCalculator", "LossCalculator", "PhotonicLanternDesignParameters"]
