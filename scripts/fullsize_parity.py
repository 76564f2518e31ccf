"""One-off parity measurement at FULL size against the oracle itself (minutes of CPU per case; the -m gpu tests check
size-independent properties there): BASELINE configs[2] rung L = 2 (N = 362 285) and configs[4] C5 (19 cores, N = 744 037).
usage: fullsize_parity.py [l2] [c5]   -> gpurun_out/fullsize_parity.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver
from oracle import hfield
from oracle.p2 import MeshTriLite
from oracle.compare import mode_field_errors

cases = {"l2": (7, 2, 10), "c5": (19, 2, 20)}
out = open("gpurun_out/fullsize_parity.txt", "a")
for name in (sys.argv[1:] or ["l2", "c5"]):
    ncore, levels, nm = cases[name]
    g = MCFGeometry(ncore, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    mesh = generate_mesh(g, 1.0, levels)
    s = TrueVectorialMaxwellSolver(g, device=0)
    s.solve_vectorial_modes(mesh, nm)
    t0 = time.perf_counter(); modes = s.solve_vectorial_modes(mesh, nm); t1 = time.perf_counter()
    st = s.last_stats
    print(name, "gpu %.1f ms (warm), N %d, k %d, residual %.2e, perturbed %d" % (1e3 * (t1 - t0), st["N"], st["n_req"], st["true_residual"], st["pivot_perturbations"]), flush=True)
    tm = {}
    ref = hfield.solve_vectorial_modes(g, MeshTriLite(mesh.p, mesh.t), n_modes_target=nm, fused=True, timings=tm)
    dn = max(abs(a["n_eff"] - b["n_eff"]) for a, b in zip(modes, ref)) if len(modes) == len(ref) else float("nan")
    fe = float(mode_field_errors(modes, ref, rel_gap=1e-6).max()) if len(modes) == len(ref) else float("nan")
    line = (f"{name}: N = {st['N']}, k = {st['n_req']}: modes {len(modes)}/{len(ref)}, max |dn_eff| = {dn:.2e}, max field L2 = {fe:.2e}; "
            f"GPU warm solve {1e3 * (t1 - t0):.1f} ms, oracle {tm['total']:.0f} s (assembly {tm['assembly']:.0f}, eigsh {tm['eigsh']:.0f})")
    print(line, flush=True)
    out.write(line + "\n"); out.flush()
    s.clear_cache()
