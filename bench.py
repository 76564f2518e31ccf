#!/usr/bin/env python3
"""Headline benchmark: eigenmodes/sec (assembly + solve) on the 7-core hexagonal P2 mesh, 10 modes.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full ``TrueVectorialMaxwellSolver.solve_vectorial_modes(mesh, 10)`` call on the
north-star cross-section C1 (BASELINE.json configs[1]: 7-core hexagonal_1plus6_7, r = 1.5 um, pitch
8 um, lambda = 1550 nm, synthetic mesh recipe at refinement 1.0 + 1 uniform refinement: N = 90 639 P2
DOFs, n = 180 742, k = 22 eigenpairs requested).  The step is COLD: a fresh solver per step, so the
mesh-only symbolic analysis (P2 numbering, CSR pattern, nested-dissection front tree), the device
context, assembly, factorisation, Lanczos, post-processing and the copy of the mode vectors back to
NumPy are all inside the timed region — everything the reference does inside the same call.  The
mesh arrays (p, t) are generated before the timed region (the reference's MeshGenerator is the step
before the path).  With N > 1 every rank solves its own independent cross-section on its own GPU
(weak scaling, no data-path collective) and the value is the aggregate over ranks.

Rank 0 prints ONE JSON line; see DESIGN.md for how ``roofline`` and ``cpu_baseline`` are measured.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_MODES = 10
METRIC = "eigenmodes/sec (assembly+solve), 7-core P2 mesh, 10 modes; |Δn_eff| vs ref"


def cpu_baseline(geom, mesh, gpu_modes):
    """Oracle (CPU port of the reference algorithm in scikit-fem's loop shape + SciPy eigsh with the
    reference's arguments) timed once on the same workload; also yields the parity numbers."""
    import numpy as np
    from oracle import hfield
    from oracle.p2 import MeshTriLite
    from oracle.compare import mode_field_errors

    threads = int(os.environ.get("PLFEM_CPU_THREADS", "4"))         # the reference sets OMP/MKL_NUM_THREADS=4 (main.py:19-20)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:                                                 # pragma: no cover
        limiter = None
    tm = {}
    t0 = time.perf_counter()
    ref = hfield.solve_vectorial_modes(geom, MeshTriLite(mesh.p, mesh.t), N_MODES, fused=False, timings=tm)
    dt = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits()
    dn = max(abs(a["n_eff"] - b["n_eff"]) for a, b in zip(gpu_modes, ref)) if len(ref) == len(gpu_modes) else float("nan")
    worst = float(np.max(mode_field_errors(gpu_modes, ref))) if len(ref) == len(gpu_modes) else float("nan")
    base = {"value": N_MODES / dt, "unit": "modes/s", "cores": threads, "kind": "port",
            "sample": f"1 full solve of the same C1 workload ({dt:.1f} s: assembly {tm['assembly']:.1f} s, "
                      f"eigsh {tm['eigsh']:.1f} s; assembly and SuperLU are single-threaded, BLAS limited to {threads} threads; "
                      f"host has {os.cpu_count()} logical CPUs)"}
    parity = {"max_abs_dn_eff": dn, "max_field_l2": worst, "n_modes_compared": len(ref),
              "tolerance": {"dn_eff": 5e-5, "field_l2": 1e-6}}
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--levels", type=int, default=1, help="uniform refinements of the synthetic mesh (1 = C1)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the eigenmode path has no CPU fallback")
    # Rehearsal of the multi-rank path on a one-GPU box (not a measurement): PLFEM_BENCH_SAME_DEVICE=1 puts every
    # rank on device 0 and PLFEM_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device.
    backend = os.environ.get("PLFEM_BENCH_BACKEND", "nccl")
    if os.environ.get("PLFEM_BENCH_SAME_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
    from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver

    geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    mesh = generate_mesh(geom, 1.0, args.levels)

    kprof = {"launches": 0, "total_us": 0.0, "bytes": 0.0}

    def step(profile=False):
        # profile: HIP events (on the launch stream) around every launch of the dominant kernel, live
        # inside the timed region -> "roofline" below.  Done for the first timed step only: the ~700
        # event records per step cost ~4 % when applied to every step.
        solver = TrueVectorialMaxwellSolver(geom, device=local_rank, reuse_symbolic=False, profile_kernel=profile)
        modes = solver.solve_vectorial_modes(mesh, N_MODES)
        kp = solver.last_stats.get("kernel_profile")
        if kp:
            for k in kprof:
                kprof[k] += kp[k]
        return solver, modes

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for w in range(args.warmup):
        # the last warm-up step also runs with the kernel timing on, so that the process-wide event pool exists
        # before the timed region (its numbers are discarded below)
        solver, modes = step(profile=(w == args.warmup - 1))
    sync()
    for k in kprof:
        kprof[k] = 0
    # per-step wall times and host phases (diagnostics only: a shared host shows up as outliers in "context")
    step_ms, host_ms = [], []
    t0 = time.perf_counter()
    for it in range(args.steps):
        ts = time.perf_counter()
        solver, modes = step(profile=(it == 0))
        step_ms.append((time.perf_counter() - ts) * 1e3)
        ls = solver.last_stats
        host_ms.append((ls["t_symbolic"] * 1e3, ls["t_context"] * 1e3, ls.get("t_workspace", 0.0) * 1e3))
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    stats = dict(solver.last_stats)

    # warm figure (symbolic analysis + context kept, e.g. the other wavelengths of a sweep) — extra info
    ws = TrueVectorialMaxwellSolver(geom, device=local_rank, reuse_symbolic=True)
    ws.solve_vectorial_modes(mesh, N_MODES)
    torch.cuda.synchronize()
    tw = time.perf_counter()
    nwarm = max(2, min(5, args.steps))
    for _ in range(nwarm):
        ws.solve_vectorial_modes(mesh, N_MODES)
    torch.cuda.synchronize()
    warm_ms = (time.perf_counter() - tw) / nwarm * 1e3
    # roofline of the dominant kernel (k_fwd: tile-form forward sweep of the shift-invert solve, HBM bound)
    roof = None
    if kprof["launches"] > 0:
        achieved = kprof["bytes"] / (kprof["total_us"] * 1e-6) / 1e9           # GB/s
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_k_fwd.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        roof = {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                "traffic": traffic, "kernel": "k_fwd<4, 4>", "launches": kprof["launches"],
                "avg_launch_us": kprof["total_us"] / kprof["launches"],
                "algorithmic_bytes_per_launch": kprof["bytes"] / kprof["launches"]}
    ws.clear_cache()

    if rank == 0:
        out = {
            "metric": METRIC, "value": world * args.steps * N_MODES / elapsed, "unit": "modes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"C1: 7-core hexagonal_1plus6_7, r=1.5um, pitch=8um, lambda=1550nm, synthetic mesh "
                                   f"recipe refinement 1.0 + {args.levels} uniform refinement(s), N={stats['N']} P2 DOFs, "
                                   f"n={stats['n']}, 10 modes requested (k={stats['n_req']}, ncv={stats['ncv']})",
                       "step": "cold solve_vectorial_modes (symbolic + context + assembly + factor + Lanczos + post + D2H)",
                       "parallelism": f"{world} independent cross-sections, one per GPU"},
            "breakdown_ms": {"symbolic_host": stats["t_symbolic"] * 1e3, "context": stats["t_context"] * 1e3,
                             "assemble": stats["assemble_us"] / 1e3, "factor": stats["factor_us"] / 1e3,
                             "lanczos": stats["lanczos_us"] / 1e3, "post": stats["post_us"] / 1e3,
                             "copy_out": stats["t_copy_out"] * 1e3, "warm_step": warm_ms},
            "lanczos": {"n_opinv": stats["n_opinv"], "restarts": stats["restarts"], "nconv": stats["nconv"]},
            "raw_eigenpairs_per_s": world * args.steps * stats["n_req"] / elapsed,
            "step_ms": [round(v, 2) for v in step_ms],
            "host_ms_max": {"symbolic": round(max(h[0] for h in host_ms), 2), "context": round(max(h[1] for h in host_ms), 2),
                            "workspace_alloc": round(max(h[2] for h in host_ms), 2)},
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            base, parity = cpu_baseline(geom, mesh, modes)
            out["cpu_baseline"] = base
            out["parity"] = parity
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
