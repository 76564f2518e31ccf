"""Field / n_eff agreement with the oracle (scikit-fem-shaped assembly + eigsh(tol=1e-7) AND eigsh(tol=1e-13)) against
the Lanczos tolerance of the device path, on the L = 0 and L = 1 rungs of the 7-core cross-section."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging; logging.disable(logging.WARNING)
import numpy as np
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver
from oracle.p2 import MeshTriLite
from oracle import hfield
from oracle.compare import mode_field_errors

geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
for levels in (0, 1):
    mesh = generate_mesh(geom, 1.0, levels)
    refs = {}
    for rt in (1e-7, 1e-13):
        t0 = time.perf_counter()
        refs[rt] = hfield.solve_vectorial_modes(geom, MeshTriLite(mesh.p, mesh.t), n_modes_target=10, fused=True, tol=rt)
        print(f"L={levels} oracle eigsh tol {rt:g}: {time.perf_counter() - t0:.1f} s", flush=True)
    print("   oracle 1e-7 vs 1e-13: field", f"{mode_field_errors(refs[1e-7], refs[1e-13], rel_gap=1e-5).max():.2e}")
    for tol in (1e-10, 1e-9, 1e-8, 1e-7):
        s = TrueVectorialMaxwellSolver(geom, device=0, eig_tol=tol)
        modes = s.solve_vectorial_modes(mesh, 10)
        st = s.last_stats
        line = f"   device tol {tol:g}: OP {st['n_opinv']} lanczos {st['lanczos_us'] / 1e3:.2f} ms true res {st['true_residual']:.1e}"
        for rt in (1e-7, 1e-13):
            ref = refs[rt]
            dn = max(abs(a["n_eff"] - b["n_eff"]) for a, b in zip(modes, ref))
            line += f" | vs eigsh({rt:g}): dn {dn:.1e} field {mode_field_errors(modes, ref, rel_gap=1e-5).max():.2e}"
        print(line, flush=True)
