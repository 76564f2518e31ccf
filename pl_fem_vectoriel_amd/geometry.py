"""Cross-section geometry for the vectorial H-field eigenmode path (host side).

Mirrors the part of the reference geometry that the hot path touches
(reference ``geometry_unified.py``):

* ``mcf_positions``            -> ``geometry_unified.py:74-188``   (12 published layouts)
* ``MCFGeometry``              -> ``geometry_unified.py:195-347``  (k0, V, radii, domain, eps(x,y), hash)
* ``PhotonicLanternGeometry``  -> ``geometry_unified.py:637-678``  (explicit positions / radii)

Only the attributes the solver, the mesh recipe and the loss consumer read are kept
(``geometry_unified.py:15-32``).  The taper / MMF / PhotonicLantern assembly classes
(``geometry_unified.py:423-630``) are outside the eigenmode path and are not provided.

Values are pinned by ``tests/golden/geometry_golden.json`` which was produced by importing the
reference module (``tests/golden/make_geometry_golden.py``).
"""
from __future__ import annotations

import hashlib
from typing import Optional, Tuple

import numpy as np

# reference geometry_unified.py:61-67
N_AIR = 1.0
PML_STRENGTH = 3.0
PML_ORDER = 2
PML_THICKNESS_UM = 10.0

SUPPORTED_N = (1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 19)

# arrangement string (= reference ``config_type``) -> (n_cores, variant)
ARRANGEMENTS = {
    "single_1": (1, None),
    "linear_2": (2, None),
    "triangular_3": (3, None),
    "square_2x2_4": (4, None),
    "pentagonal_ring_5": (5, None),
    "hexagonal_ring_6": (6, None),
    "pentagon_center_6": (6, "pentagon_center"),
    "hexagonal_1plus6_7": (7, None),
    "heptagonal_center_8": (8, None),
    "square_3x3_9": (9, None),
    "hex_double_ring_12": (12, None),
    "hex_1plus6plus6_13": (13, None),
    "hex_1plus6plus12_19": (19, None),
}


def _ring(radius: float, angles_deg) -> np.ndarray:
    a = np.radians(angles_deg)
    return radius * np.column_stack([np.cos(a), np.sin(a)])


def mcf_positions(n_cores: int, pitch: float, variant: Optional[str] = None
                  ) -> Tuple[np.ndarray, str, bool, int, float]:
    """Core centres for the published multi-core layouts (``geometry_unified.py:74-188``).

    Returns ``(positions (N,2), config_type, has_central_core, n_peripheral, R_ring)``.
    """
    p = float(pitch)
    hex6 = np.arange(6) * 60
    if n_cores == 1:
        return np.array([[0.0, 0.0]]), "single_1", True, 0, 0.0
    if n_cores == 2:
        return np.array([[-p / 2, 0.0], [p / 2, 0.0]]), "linear_2", False, 2, p / 2
    if n_cores == 3:
        return _ring(p, [90, 210, 330]), "triangular_3", False, 3, p
    if n_cores == 4:
        h = p / 2
        return (np.array([[-h, -h], [h, -h], [-h, h], [h, h]]),
                "square_2x2_4", False, 4, h * np.sqrt(2))
    if n_cores == 5:
        return _ring(p, 90 + np.arange(5) * 72), "pentagonal_ring_5", False, 5, p
    if n_cores == 6:
        if variant == "pentagon_center":
            ring = _ring(p, 90 + np.arange(5) * 72)
            return np.vstack([[0.0, 0.0], ring]), "pentagon_center_6", True, 5, p
        return _ring(p, hex6), "hexagonal_ring_6", False, 6, p
    if n_cores == 7:
        return np.vstack([[0.0, 0.0], _ring(p, hex6)]), "hexagonal_1plus6_7", True, 6, p
    if n_cores == 8:
        ring = _ring(p, np.arange(7) * (360 / 7))
        return np.vstack([[0.0, 0.0], ring]), "heptagonal_center_8", True, 7, p
    if n_cores == 9:
        c = [-p, 0.0, p]
        pos = np.array([[x, y] for y in c for x in c])
        return pos, "square_3x3_9", True, 8, p * np.sqrt(2)
    if n_cores == 12:
        return (np.vstack([_ring(p, hex6), _ring(p * np.sqrt(3), hex6 + 30)]),
                "hex_double_ring_12", False, 12, p * np.sqrt(3))
    if n_cores == 13:
        return (np.vstack([[0.0, 0.0], _ring(p, hex6), _ring(p * np.sqrt(3), hex6 + 30)]),
                "hex_1plus6plus6_13", True, 12, p * np.sqrt(3))
    if n_cores == 19:
        a1 = np.radians(hex6)
        a2 = np.radians(hex6 + 30)
        pos = [[0.0, 0.0]]
        pos += [[p * np.cos(a), p * np.sin(a)] for a in a1]
        pos += [[2 * p * np.cos(a), 2 * p * np.sin(a)] for a in a1]
        pos += [[p * np.sqrt(3) * np.cos(a), p * np.sqrt(3) * np.sin(a)] for a in a2]
        return np.array(pos), "hex_1plus6plus12_19", True, 18, 2 * p
    raise ValueError(f"n_cores={n_cores} non supporté. Valides : {list(SUPPORTED_N)}")


class MCFGeometry:
    """Multi-core fibre cross-section (``geometry_unified.py:195-347``)."""

    SUPPORTED_N = list(SUPPORTED_N)

    def __init__(self, n_cores: int, pitch_um: float, core_radius_um: float, n_core: float,
                 n_clad: float = N_AIR, wavelength_um: float = 1.55,
                 cladding_radius: Optional[float] = None,
                 pml_thickness: float = PML_THICKNESS_UM, pml_strength: float = PML_STRENGTH,
                 pml_order: int = PML_ORDER, use_complex_pml: bool = True,
                 taper_length_um: Optional[float] = None, variant: Optional[str] = None):
        self.n_cores = int(n_cores)
        self.n_core = float(n_core)
        self.n_clad = float(n_clad)
        self.delta_n = self.n_core - self.n_clad
        self.wavelength = float(wavelength_um)
        self.k0 = 2 * np.pi / self.wavelength                       # :232
        if self.delta_n < 1e-6:                                      # :234-235
            raise ValueError(f"Δn={self.delta_n:.2e} trop faible")

        (self.positions, self.config_type, self.has_central_core,
         self.n_peripheral, self.R_ring) = mcf_positions(n_cores, pitch_um, variant)
        self.core_radii = np.full(self.n_cores, float(core_radius_um))
        self.core_positions = self.positions
        self.r_core = float(core_radius_um)
        self.V_number = self.k0 * self.r_core * np.sqrt(max(self.n_core ** 2 - self.n_clad ** 2, 0.0))

        if n_cores > 1:                                              # :255-264
            pos = self.positions
            d = [np.linalg.norm(pos[i] - pos[j]) for i in range(n_cores) for j in range(i + 1, n_cores)]
            self.pitch = self.pitch_min = float(np.min(d))
            max_r = float(np.max(np.linalg.norm(pos, axis=1)))
        else:
            self.pitch = self.pitch_min = 0.0
            max_r = 0.0
        self.pitch_ratio = self.pitch / (2 * self.r_core) if self.r_core > 0 else 0.0

        self.cladding_radius = (cladding_radius if cladding_radius is not None
                                else max(max_r * 1.8 + self.r_core * 2, 20.0))       # :269-272
        self._domain_radius = max(max_r + self.r_core * 4,
                                  self.cladding_radius + pml_thickness * 1.2)        # :275-278
        self.pml_thickness = float(pml_thickness)
        self.pml_strength = float(pml_strength)
        self.pml_order = int(pml_order)
        self.use_complex_pml = bool(use_complex_pml)
        self.taper_length = taper_length_um

        area_c = n_cores * np.pi * self.r_core ** 2
        area_t = np.pi * (max_r + self.r_core) ** 2 if n_cores > 1 else area_c
        self.packing_efficiency = float(area_c / max(area_t, 1e-9))
        self._hash = self._compute_hash()

    # -- properties ---------------------------------------------------------------------------
    @property
    def domain_radius(self) -> float:
        return self._domain_radius

    @property
    def hash(self) -> str:
        return self._hash

    def _compute_hash(self) -> str:                                  # :313-321
        h = hashlib.sha256()
        h.update(str(self.n_cores).encode())
        h.update(self.positions.tobytes())
        h.update(self.core_radii.tobytes())
        h.update(f"{self.n_core:.6f}{self.n_clad:.6f}{self.wavelength:.6f}".encode())
        h.update(f"{self.cladding_radius:.4f}{self.pml_thickness:.2f}".encode())
        h.update(str(self.use_complex_pml).encode())
        return h.hexdigest()[:20]

    # -- eps(x, y) ----------------------------------------------------------------------------
    def epsilon(self, x, y) -> np.ndarray:
        """Complex relative permittivity (``geometry_unified.py:325-347``).

        Closed core discs (later cores overwrite earlier ones) on an ``n_clad**2`` background,
        times ``1 + i s rho**order`` in the annular PML.  The eigenmode path only reads the real
        part (``solver_fem.py:132-150``) so the PML never changes A or B.
        """
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        eps = np.full_like(x, self.n_clad ** 2, dtype=np.complex128)
        for (cx, cy), r in zip(self.positions, self.core_radii):
            eps[(x - cx) ** 2 + (y - cy) ** 2 <= r ** 2] = self.n_core ** 2
        if self.use_complex_pml:
            r_dist = np.sqrt(x ** 2 + y ** 2)
            start = self._domain_radius - self.pml_thickness
            m = r_dist > start
            if np.any(m):
                rn = np.clip((r_dist[m] - start) / self.pml_thickness, 0.0, 1.0)
                eps[m] *= (1.0 + 1j * self.pml_strength * rn ** self.pml_order)
        return eps

    def core_table(self) -> np.ndarray:
        """``(n_cores, 3)`` float64 rows ``cx, cy, r`` — the layout the HIP kernels take."""
        return np.ascontiguousarray(
            np.column_stack([np.asarray(self.positions, dtype=np.float64).reshape(-1, 2),
                             np.asarray(self.core_radii, dtype=np.float64)]))

    def validate(self) -> Tuple[bool, str]:                          # :351-363
        if self.delta_n < 5e-4:
            return False, f"Δn trop faible ({self.delta_n:.2e})"
        if self.V_number < 0.5:
            return False, f"V-number trop faible ({self.V_number:.2f})"
        if self.V_number > 20.0:
            return False, f"V-number très élevé ({self.V_number:.2f}) → multimode"
        for i in range(self.n_cores):
            for j in range(i + 1, self.n_cores):
                d = np.linalg.norm(self.positions[i] - self.positions[j])
                if d < (self.core_radii[i] + self.core_radii[j]) * 0.85:
                    return False, f"Chevauchement cœurs {i}↔{j}: d={d:.2f}µm"
        return True, "OK"

    def __repr__(self) -> str:
        return (f"MCFGeometry(N={self.n_cores}, {self.config_type}, pitch={self.pitch:.1f}µm, "
                f"r={self.r_core:.2f}µm, V={self.V_number:.2f}, n={self.n_core:.4f}/{self.n_clad:.4f})")


class PhotonicLanternGeometry(MCFGeometry):
    """Explicit-positions geometry (``geometry_unified.py:637-678``).

    Two call forms are accepted:

    * the code form ``PhotonicLanternGeometry(n_cores, arrangement, core_positions, core_radii,
      n_core, n_clad=1.0, cladding_radius=None, wavelength=1.55[µm], ...)``;
    * the documented form (reference ``README.md:141-148``)
      ``PhotonicLanternGeometry(arrangement="hexagonal_1plus6_7", core_radius_um=, pitch_um=,
      n_core=, n_clad=, wavelength_nm=)`` which builds positions from ``mcf_positions``.
    """

    def __init__(self, n_cores=None, arrangement=None, core_positions=None, core_radii=None,
                 n_core=None, n_clad=1.0, cladding_radius=None, wavelength=1.55, taper_length=None,
                 pml_thickness=10.0, pml_strength=3.0, pml_order=2, use_complex_pml=True, **kwargs):
        if core_positions is None:
            # documented keyword form
            if arrangement is None and isinstance(n_cores, str):
                arrangement, n_cores = n_cores, None
            if arrangement not in ARRANGEMENTS:
                raise ValueError(f"arrangement inconnu: {arrangement!r}; valides: {sorted(ARRANGEMENTS)}")
            nc, variant = ARRANGEMENTS[arrangement]
            pitch_um = float(kwargs.pop("pitch_um"))
            r_um = float(kwargs.pop("core_radius_um"))
            if "wavelength_nm" in kwargs:
                wavelength = float(kwargs.pop("wavelength_nm")) * 1e-3
            if n_core is None:
                raise ValueError("n_core requis")
            core_positions = mcf_positions(nc, pitch_um, variant)[0]
            core_radii = np.full(nc, r_um)
            n_cores = nc
        positions = np.atleast_2d(np.asarray(core_positions, dtype=np.float64))
        if len(positions) > 1:
            d = [np.linalg.norm(positions[i] - positions[j])
                 for i in range(len(positions)) for j in range(i + 1, len(positions))]
            pitch = float(np.min(d))
        else:
            pitch = float(np.max(core_radii)) * 4
        r_core = float(np.mean(core_radii))
        super().__init__(n_cores=n_cores, pitch_um=pitch, core_radius_um=r_core, n_core=n_core,
                         n_clad=n_clad, wavelength_um=wavelength, cladding_radius=cladding_radius,
                         pml_thickness=pml_thickness, pml_strength=pml_strength, pml_order=pml_order,
                         use_complex_pml=use_complex_pml, taper_length_um=taper_length)
        self.positions = positions
        self.core_positions = positions
        self.core_radii = np.asarray(core_radii, dtype=np.float64)
        self.arrangement = str(arrangement)
