"""Wider GPU-vs-oracle regression than the pytest suite (minutes of oracle time): several arrangements, mesh sizes
and mode counts; prints max |dn_eff|, field error, Lanczos statistics and the factorisation's perturbation count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
from pl_fem_vectoriel_amd.geometry import ARRANGEMENTS
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver
from oracle import hfield
from oracle.p2 import MeshTriLite
from oracle.compare import mode_field_errors

cases = [("hexagonal_1plus6_7", 1.0, 0, 10, 1.55), ("hexagonal_1plus6_7", 1.0, 1, 30, 1.31), ("square_2x2_4", 0.7, 1, 10, 1.55),
         ("hex_1plus6plus12_19", 0.6, 1, 20, 1.55), ("single_1", 1.0, 1, 4, 1.65), ("hex_double_ring_12", 0.5, 1, 50, 1.49)]
worst = 0.0
for arr, ref_, lv, nm, lam in cases:
    n, variant = ARRANGEMENTS[arr]
    g = MCFGeometry(n, 8.0, 1.5, 1.535, 1.0, wavelength_um=lam, variant=variant)
    mesh = generate_mesh(g, ref_, lv)
    s = TrueVectorialMaxwellSolver(g, device=0)
    t0 = time.perf_counter(); modes = s.solve_vectorial_modes(mesh, nm); t1 = time.perf_counter()
    ref = hfield.solve_vectorial_modes(g, MeshTriLite(mesh.p, mesh.t), n_modes_target=nm, fused=True)
    t2 = time.perf_counter()
    st = s.last_stats
    ok = len(modes) == len(ref)
    dn = max((abs(a["n_eff"] - b["n_eff"]) for a, b in zip(modes, ref)), default=0.0) if ok else float("nan")
    fe = float(mode_field_errors(modes, ref, rel_gap=1e-5).max()) if ok and modes else 0.0
    worst = max(worst, dn if ok else 1.0)
    print(f"{arr:22s} ref={ref_} L{lv} k={st['n_req']:3d} N={st['N']:7d}  gpu {1e3*(t1-t0):7.1f} ms  oracle {t2-t1:6.1f} s  modes {len(modes)}/{len(ref)}  "
          f"|dn_eff| {dn:.1e}  field {fe:.1e}  solves {st.get('n_block_solves', st['n_opinv'])} restarts {st['restarts']} pert {st['pivot_perturbations']}", flush=True)
print("worst |dn_eff|", worst)
sys.exit(0 if worst < 5e-5 else 1)
