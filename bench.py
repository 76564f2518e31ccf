#!/usr/bin/env python3
"""Headline benchmark: eigenmodes/sec (assembly + solve) on the 7-core hexagonal P2 mesh, 10 modes.

    python bench.py --gpus N --steps K --warmup W [--levels L | --ladder | --sweep]

One "step" = one full ``TrueVectorialMaxwellSolver.solve_vectorial_modes(mesh, 10)`` call on the
north-star cross-section C1 (BASELINE.json configs[1]: 7-core hexagonal_1plus6_7, r = 1.5 um, pitch
8 um, lambda = 1550 nm, synthetic mesh recipe at refinement 1.0 + 1 uniform refinement: N = 90 639 P2
DOFs, n = 180 742, k = 22 eigenpairs requested).  The step is COLD: a fresh solver per step, so the
mesh-only symbolic analysis (P2 numbering, CSR pattern, nested-dissection front tree), the device
context, assembly, factorisation, Lanczos, the a-posteriori residual check, post-processing and the
copy of the mode vectors back to NumPy are all inside the timed region — everything the reference does
inside the same call.  The mesh arrays (p, t) are generated before the timed region (the reference's
MeshGenerator is the step before the path).

``--gpus N`` with N > 1: when the process is not already a rank (no RANK in the environment) it starts N
ranks of itself — fresh child processes, before anything touches the GPU — and relays rank 0's line;
under ``python -m torch.distributed.run`` it is one of the ranks.  Every rank solves its own independent
cross-section on its own GPU (weak scaling, no data-path collective), the value is
the aggregate over ranks.  ``--sweep``: a step is BASELINE.json configs[3] instead, the 64-solve multi-band
sweep sharded over the ranks with one RCCL all-gather of the result records per step (strong scaling).
``--ladder``: one line per rung of BASELINE.json configs[2] (L = 0, 1, 2 uniform refinements).

Rank 0 prints ONE JSON line per configuration; see DESIGN.md for how ``roofline`` and ``cpu_baseline`` are measured.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_MODES = 10
SWEEP_LANES = 4      # --sweep: solves in flight per GPU unless --lanes says otherwise (1-GPU sweep: 1.42 s with 1, 0.89 s with 4)
METRIC = "eigenmodes/sec (assembly+solve), 7-core P2 mesh, 10 modes; |Δn_eff| vs ref"
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_MFMA_PEAK_TF = 78.6       # v_mfma_f64_16x16x4_f64: 256 CUs x 4 SIMDs x 32 flop/clk x 2.4 GHz


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--levels", type=int, default=1, help="uniform refinements of the synthetic mesh (1 = C1)")
    ap.add_argument("--ladder", action="store_true", help="BASELINE configs[2]: one line per rung L = 0, 1, 2")
    ap.add_argument("--sweep", action="store_true", help="BASELINE configs[3]: a step = the 64-solve multi-band sweep")
    ap.add_argument("--lanes", type=int, default=0, help="--sweep: solves in flight per GPU (0 = the driver's default)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` outside torchrun
# ----------------------------------------------------------------------------------------------------
def launch_ranks(args, argv) -> int:
    """Start ``args.gpus`` ranks of this script as fresh child processes (the parent has not imported torch or
    touched the GPU, and never re-executes itself), wait for them, relay rank 0's output.  Returns the exit code."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL       # only rank 0 prints the line; stderr is shared
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out))
    # rank 0's stdout is small (a few JSON lines): read it to the end first, then reap; if any rank dies early the
    # others would wait in a collective forever, so poll and stop the exact children that were started here
    failed = None
    pending = set(range(args.gpus))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and failed is None:
                failed = (r, rc)
        if failed is not None and pending:
            time.sleep(5.0)                                    # let the others fail on their own (collective error) first
            for r in sorted(pending):
                if procs[r].poll() is None:
                    procs[r].terminate()
            for r in sorted(pending):
                try:
                    procs[r].wait(timeout=30)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            pending.clear()
        time.sleep(0.05)
    reader.join(timeout=10)
    text = (chunks[0] if chunks else b"").decode()
    sys.stdout.write(text)
    sys.stdout.flush()
    if failed is not None:
        sys.stderr.write(f"bench.py: rank {failed[0]} exited with code {failed[1]}\n")
        return 1
    return 0


# ----------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only)
# ----------------------------------------------------------------------------------------------------
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
            "sample": f"1 full solve of the same workload ({dt:.1f} s: assembly {tm['assembly']:.1f} s, "
                      f"eigsh {tm['eigsh']:.1f} s; assembly and SuperLU are single-threaded, BLAS limited to {threads} threads; "
                      f"host has {os.cpu_count()} logical CPUs)"}
    parity = {"max_abs_dn_eff": dn, "max_field_l2": worst, "n_modes_compared": len(ref),
              "tolerance": {"dn_eff": 5e-5, "field_l2": 1e-6}}
    return base, parity


# ----------------------------------------------------------------------------------------------------
# rooflines
# ----------------------------------------------------------------------------------------------------
def roofline_objects(kprof, stats, info, nv, ne):
    """``roofline`` (dominant kernel, k_fwd) and ``roofline.kernels`` (the other kernel families of the step).
    achieved = ALGORITHMIC bytes (or flop) / measured time; formulas in DESIGN.md section 6.
    kprof: HIP-event ranges accumulated over the profiled step (plfem_profile_*); stats: solver.last_stats of that
    step (assemble_us / factor_us are HIP-event times of the whole phase); info: plfem_symbolic_info."""
    def hbm(name, bytes_, us, n, formula):
        a = bytes_ / (us * 1e-6) / 1e9
        return {"kernel": name, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                "ranges_timed": n, "avg_us": us / n, "algorithmic_bytes": bytes_ / n, "formula": formula}

    sl = kprof["slots"]
    kernels = []
    sweep_formula = "8 B x sum over fronts (s2 m - s2^2/2 + P (m + s2)), P = 4"
    for key, name in (("fwd_sweep", "forward sweep (all levels: k_fwd + k_fwd_rows)"),
                      ("bwd_sweep", "backward sweep (all levels: k_bwd + k_bwd_rows)")):
        if sl[key]["ranges"] > 0:
            kernels.append(hbm(name, sl[key]["bytes"], sl[key]["total_us"], sl[key]["ranges"], sweep_formula))
    if sl["spmv_b"]["ranges"] > 0:
        kernels.append(hbm("k_spmv_b_block<4>", sl["spmv_b"]["bytes"], sl["spmv_b"]["total_us"], sl["spmv_b"]["ranges"],
                           "12 B x nnz + 4 B x (N + 1) + 2 x 8 B x 4 x 2N"))
    nnz = info["nnz"]
    asm_bytes = 24.0 * ne + 16.0 * nv + 8.0 * 5 * nnz
    if stats.get("assemble_us", 0) > 0:
        kernels.append(hbm("assembly (k_element_matrices + k_csr_gather)", asm_bytes, stats["assemble_us"], 1,
                           "24 B x ne (t, edge dofs) + 16 B x nv (coordinates) + 8 B x 5 x nnz (Axx Axy Ayx Ayy Minv values written once)"))
    if stats.get("factor_us", 0) > 0:
        tf = info["factor_flops"] / (stats["factor_us"] * 1e-6) / 1e12
        kernels.append({"kernel": "factorisation (block LDL^T, all kernels)", "bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TF,
                        "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TF, "ranges_timed": 1, "avg_us": stats["factor_us"],
                        "algorithmic_flop": info["factor_flops"], "formula": "sum over fronts s2 m^2 (LDL^T + Schur complement + L11^-1 + Z)"})
    roof = None
    if kprof["launches"] > 0:
        achieved = kprof["bytes"] / (kprof["total_us"] * 1e-6) / 1e9
        traffic, traffic_src = None, None
        for tag in ("r02", "r01"):          # PMC passes run separately (scripts/gpu_profile_round.sh); newest committed file
            pmc = os.path.join(ROOT, "profiles", f"{tag}_pmc_k_fwd.json")
            if os.path.exists(pmc):
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                traffic_src = f"profiles/{tag}_pmc_k_fwd.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of that round, not this run)"
                break
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_src, "kernel": "k_fwd<4> / k_fwd_mix<4> (tile-form forward sweep levels)",
                "launches": kprof["launches"],
                "avg_launch_us": kprof["total_us"] / kprof["launches"],
                "algorithmic_bytes_per_launch": kprof["bytes"] / kprof["launches"],
                "kernels": kernels}
    return roof


# ----------------------------------------------------------------------------------------------------
# one configuration = one JSON line
# ----------------------------------------------------------------------------------------------------
class Dist:
    """Rank bookkeeping + barrier / max-over-ranks for both backends (and the GPU-less rehearsal)."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        # Rehearsal of the multi-rank path without N GPUs (not a measurement): PLFEM_BENCH_BACKEND=gloo replaces
        # RCCL, PLFEM_BENCH_SAME_DEVICE=1 puts every rank on device 0, PLFEM_BENCH_FAKE=1 replaces the solve by a
        # sleep so that the launcher / barrier / reduction logic runs on a box with no GPU at all (tests/).
        self.backend = os.environ.get("PLFEM_BENCH_BACKEND", "nccl")
        self.fake = bool(os.environ.get("PLFEM_BENCH_FAKE"))
        if self.world != args.gpus:
            raise RuntimeError(f"--gpus {args.gpus} but WORLD_SIZE={self.world}: launch with `python bench.py --gpus N` "
                               f"or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
        import torch
        self.torch = torch
        if not self.fake:
            if not torch.cuda.is_available():
                raise RuntimeError("bench.py needs a GPU: the eigenmode path has no CPU fallback")
            if os.environ.get("PLFEM_BENCH_SAME_DEVICE"):
                self.local_rank = 0
            torch.cuda.set_device(self.local_rank)
        self.dist = None
        if self.world > 1:
            import torch.distributed as dist
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group(self.backend)
            assert dist.get_world_size() == args.gpus
            self.dist = dist

    def sync(self):
        if not self.fake:
            self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            if not self.fake:
                self.torch.cuda.synchronize()

    def max_over_ranks(self, x: float) -> float:
        if self.dist is None:
            return x
        dev = "cuda" if (self.backend == "nccl" and not self.fake) else "cpu"
        t = self.torch.tensor([x], dtype=self.torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def run_solve_config(args, D: Dist, levels: int, with_cpu_baseline: bool):
    """The headline configuration (cold C1 solve per step) or another rung of the ladder."""
    world = D.world
    if D.fake:
        # launcher / collective rehearsal: a step is a 5 ms sleep
        for _ in range(args.warmup):
            time.sleep(0.005)
        D.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            time.sleep(0.005)
        D.sync()
        elapsed = D.max_over_ranks(time.perf_counter() - t0)
        return {"metric": METRIC, "value": world * args.steps * N_MODES / elapsed, "unit": "modes/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "fake (rehearsal)",
                "config": {"workload": "PLFEM_BENCH_FAKE rehearsal: 5 ms sleep per step"}}
    torch = D.torch
    from pl_fem_vectoriel_amd import MCFGeometry, generate_mesh
    from pl_fem_vectoriel_amd.solver_fem import TrueVectorialMaxwellSolver

    geom = MCFGeometry(7, 8.0, 1.5, 1.535, 1.0, wavelength_um=1.55)
    mesh = generate_mesh(geom, 1.0, levels)
    kprof = {"launches": 0, "total_us": 0.0, "bytes": 0.0, "slots": None}
    prof_stats = {}

    def step(profile=False):
        # profile: HIP events (on the launch stream) around every launch of the dominant kernel, around the two
        # sweeps and the block SpMV, live inside the timed region -> "roofline" below.  Done for the first timed
        # step only: the ~1000 event records per step cost a few percent when applied to every step.
        solver = TrueVectorialMaxwellSolver(geom, device=D.local_rank, reuse_symbolic=False, profile_kernel=profile)
        modes = solver.solve_vectorial_modes(mesh, N_MODES)
        kp = solver.last_stats.get("kernel_profile")
        if kp:
            for k in ("launches", "total_us", "bytes"):
                kprof[k] += kp[k]
            kprof["slots"] = kp["slots"]
            prof_stats.update(solver.last_stats)
        return solver, modes

    for w in range(args.warmup):
        # the last warm-up step also runs with the kernel timing on, so that the process-wide event pool exists
        # before the timed region (its numbers are discarded below)
        solver, modes = step(profile=(w == args.warmup - 1))
    D.sync()
    kprof.update(launches=0, total_us=0.0, bytes=0.0, slots=None)
    # per-step wall times and host phases (diagnostics only: a shared host shows up as outliers in "context")
    step_ms, host_ms = [], []
    t0 = time.perf_counter()
    for it in range(args.steps):
        ts = time.perf_counter()
        solver, modes = step(profile=(it == 0))
        step_ms.append((time.perf_counter() - ts) * 1e3)
        ls = solver.last_stats
        host_ms.append((ls["t_symbolic"] * 1e3, ls["t_context"] * 1e3, ls.get("t_workspace", 0.0) * 1e3))
    D.sync()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    stats = dict(solver.last_stats)

    # warm figure (symbolic analysis + context kept, e.g. the other wavelengths of a sweep) — extra info
    ws = TrueVectorialMaxwellSolver(geom, device=D.local_rank, reuse_symbolic=True)
    ws.solve_vectorial_modes(mesh, N_MODES)
    torch.cuda.synchronize()
    tw = time.perf_counter()
    nwarm = max(2, min(5, args.steps))
    for _ in range(nwarm):
        ws.solve_vectorial_modes(mesh, N_MODES)
    torch.cuda.synchronize()
    warm_ms = (time.perf_counter() - tw) / nwarm * 1e3
    info = next(iter(ws._cache.values()))["sym"].info
    roof = roofline_objects(kprof, prof_stats or stats, info, mesh.p.shape[1], mesh.t.shape[1]) if kprof["slots"] else None
    ws.clear_cache()
    # extra info, outside the timed region: the same cold solves with several in flight on this GPU (one host thread
    # and stream each) -- a single solve is a chain of latency-bound launches and leaves most of the GPU idle
    in_flight, per_lane = 4, 3
    conc = None
    if world == 1 and levels == 1 and with_cpu_baseline:    # (the extras of the default run; profiling runs pass --no-cpu-baseline)
        import threading

        def lane(n):
            with torch.cuda.stream(torch.cuda.Stream(device=D.local_rank)):
                for _ in range(n):
                    TrueVectorialMaxwellSolver(geom, device=D.local_rank, reuse_symbolic=False).solve_vectorial_modes(mesh, N_MODES)

        for n in (1, per_lane):                      # one round to warm the lanes' allocators, one timed
            threads = [threading.Thread(target=lane, args=(n,)) for _ in range(in_flight)]
            torch.cuda.synchronize()
            tc = time.perf_counter()
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            torch.cuda.synchronize()
            conc = in_flight * n * N_MODES / (time.perf_counter() - tc)
    if D.rank != 0:
        return None
    name = {1: "C1"}.get(levels, f"C3 ladder rung L={levels}")
    out = {
        "metric": METRIC, "value": world * args.steps * N_MODES / elapsed, "unit": "modes/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{name}: 7-core hexagonal_1plus6_7, r=1.5um, pitch=8um, lambda=1550nm, synthetic mesh "
                               f"recipe refinement 1.0 + {levels} uniform refinement(s), N={stats['N']} P2 DOFs, "
                               f"n={stats['n']}, 10 modes requested (k={stats['n_req']}, ncv={stats['ncv']})",
                   "step": "cold solve_vectorial_modes (symbolic + context + assembly + factor + Lanczos + residual check + post + D2H)",
                   "parallelism": f"{world} independent cross-sections, one per GPU"},
        "breakdown_ms": {"symbolic_host": stats["t_symbolic"] * 1e3, "context": stats["t_context"] * 1e3,
                         "assemble": stats["assemble_us"] / 1e3, "factor": stats["factor_us"] / 1e3,
                         "lanczos": stats["lanczos_us"] / 1e3, "post": stats["post_us"] / 1e3,
                         "copy_out": stats["t_copy_out"] * 1e3, "warm_step": warm_ms},
        "lanczos": {"n_opinv": stats["n_opinv"], "restarts": stats["restarts"], "nconv": stats["nconv"],
                    "true_residual": stats.get("true_residual"), "refined": stats.get("refined")},
        "raw_eigenpairs_per_s": world * args.steps * stats["n_req"] / elapsed,
        "step_ms": [round(v, 2) for v in step_ms],
        "host_ms_max": {"symbolic": round(max(h[0] for h in host_ms), 2), "context": round(max(h[1] for h in host_ms), 2),
                        "workspace_alloc": round(max(h[2] for h in host_ms), 2)},
        "roofline": roof,
    }
    if conc is not None:
        out["concurrent"] = {"in_flight": in_flight, "value": conc, "unit": "modes/s",
                             "note": "same cold solves, several in flight on the one GPU (host thread + stream each); "
                                     "not the headline: the reference runs one solve at a time"}
    if world == 1 and with_cpu_baseline:
        base, parity = cpu_baseline(geom, mesh, modes)
        out["cpu_baseline"] = base
        out["parity"] = parity
    return out


def run_sweep_config(args, D: Dist):
    """BASELINE.json configs[3]: a step = the 64 (arrangement x wavelength) solves, sharded over the ranks by
    pl_fem_vectoriel_amd.sweep (4 wavelengths of a mesh on one rank), one all-gather of the records per step."""
    import numpy as np
    from pl_fem_vectoriel_amd.sweep import multiband_sweep_items, partition, run_sweep, default_solve
    items = multiband_sweep_items(n_modes=N_MODES)
    world = D.world
    if D.fake:
        def solve(item, cache):
            time.sleep(0.01)
            return 1.26 + 1e-3 * item.index - 1e-5 * np.arange(4)
        device = None
    else:
        from pl_fem_vectoriel_amd.mesh import generate_mesh
        mine = partition(items, world)[D.rank]
        meshes = {}
        for it in mine:                                   # mesh producer = the step before the path: outside the timed region
            if it.mesh_key not in meshes:
                meshes[it.mesh_key] = generate_mesh(it.geometry(), it.mesh_refinement, it.mesh_levels)
        solve = default_solve(D.local_rank, meshes=meshes)
        device = D.local_rank
    table = None
    lanes = args.lanes if args.lanes > 0 else SWEEP_LANES
    for _ in range(args.warmup):
        table, _n = run_sweep(items, D.rank, world, solve=solve, device=device, lanes=lanes)
    D.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        table, _n = run_sweep(items, D.rank, world, solve=solve, device=device, lanes=lanes)
    D.sync()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    if D.rank != 0:
        return None
    assert sorted(table) == list(range(len(items)))
    return {"metric": METRIC, "value": args.steps * len(items) * N_MODES / elapsed, "unit": "modes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "fake (rehearsal)" if D.fake else "synthetic",
            "config": {"workload": "C4: 64-solve multi-band sweep = 16 cross-sections (12 multi-core layouts at pitch 8 um + 7-core at "
                                   "pitch 6/7/9/10 um) x lambda in {1490, 1550, 1600, 1650} nm, 10 modes each, meshes as C1",
                       "step": "one whole sweep: per mesh symbolic + context once, per wavelength assembly + factor + Lanczos + "
                               "check + post; one all-gather of 64 fixed-size records",
                       "parallelism": f"{world} rank(s), {len(items) // world} solves per GPU, {lanes} in flight per GPU, "
                                      f"backend {D.backend}"},
            "sweep": {"solves": len(items), "solves_per_s": args.steps * len(items) / elapsed,
                      "n_eff_checksum": float(sum(float(np.sum(table[i])) for i in sorted(table)))}}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args, argv)
    D = Dist(args)
    lines = []
    if args.sweep:
        lines.append(run_sweep_config(args, D))
    elif args.ladder:
        for L in (0, 1, 2):
            lines.append(run_solve_config(args, D, L, with_cpu_baseline=False))
    else:
        lines.append(run_solve_config(args, D, args.levels, with_cpu_baseline=not args.no_cpu_baseline))
    if D.rank == 0:
        for out in lines:
            print(json.dumps(out), flush=True)
    D.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
