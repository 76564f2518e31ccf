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
# the library no longer re-tunes the host allocator on its own: this application opts in
os.environ.setdefault("PLFEM_MALLOC_TUNE", "1")

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
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time the CPU baseline with BLAS unrestricted")
    ap.add_argument("--steady-seconds", type=float, default=6.0,
                    help="extra info at N = 1: the same cold solves repeated for this long after the CPU leg (0 = skip)")
    ap.add_argument("--levels", type=int, default=1, help="uniform refinements of the synthetic mesh (1 = C1)")
    ap.add_argument("--ladder", action="store_true", help="BASELINE configs[2]: one line per rung L = 0, 1, 2")
    ap.add_argument("--sweep", action="store_true", help="BASELINE configs[3]: a step = the 64-solve multi-band sweep")
    ap.add_argument("--lanes", type=int, default=0, help="--sweep: solves in flight per GPU (0 = the driver's default)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` outside torchrun
# ----------------------------------------------------------------------------------------------------
RENDEZVOUS_ERRORS = ("EADDRINUSE", "Address already in use", "address already in use", "DistNetworkError",
                     "failed to bind", "Connection refused", "TCPStore")


def is_rendezvous_failure(rc: int, rank0_stdout: str, stderr: str) -> bool:
    """A rank that EXITED (not: was killed by a signal) before rank 0 printed anything, with a c10d store / bind / connect
    error in its stderr: the one start-up failure worth a second launch on a fresh port."""
    return rc > 0 and not rank0_stdout.strip() and any(m in stderr for m in RENDEZVOUS_ERRORS)


def launch_ranks(args, argv) -> int:
    """Start ``args.gpus`` ranks of this script as fresh child processes (the parent has not imported torch or
    touched the GPU, and never re-executes itself), wait for them, relay rank 0's output.  Returns the exit code.
    A rendezvous port found free here can be taken by somebody else before rank 0 listens on it: ONLY when a rank fails
    with evidence of that in its stderr (``RENDEZVOUS_ERRORS``: address in use, c10d store / connect errors) is the launch
    repeated once on a fresh port.  Any other start-up failure -- an import error, out of memory, a HIP error, a rank
    killed by a signal -- is final: non-zero exit code and that rank's stderr tail.  ``PLFEM_BENCH_TIMEOUT``
    (seconds, default 3600) bounds the whole launch: ranks still running then are stopped and the exit code is non-zero."""
    limit = float(os.environ.get("PLFEM_BENCH_TIMEOUT", "3600"))
    for attempt in range(2):
        rc, retry = _launch_once(args, argv, limit)
        if rc == 0 or not retry or attempt == 1:
            return rc
        sys.stderr.write("bench.py: the rendezvous port was taken before rank 0 could listen on it, retrying once on a new port\n")
    return rc


def _launch_once(args, argv, limit: float):
    import socket
    import tempfile
    import threading

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs, errs = [], []
    t_start = time.time()
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL       # only rank 0 prints the line
        errs.append(tempfile.TemporaryFile())                         # every rank's stderr is kept (and relayed below)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env, stdout=out, stderr=errs[-1]))
    # rank 0's stdout is small (a few JSON lines): read it to the end first, then reap; if any rank dies early the
    # others would wait in a collective forever, so poll and stop the exact children that were started here
    failed = None
    timed_out = False
    pending = set(range(args.gpus))
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def stop_pending():
        for r in sorted(pending):
            if procs[r].poll() is None:
                procs[r].terminate()
        for r in sorted(pending):
            try:
                procs[r].wait(timeout=30)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        pending.clear()

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
            stop_pending()
        elif pending and time.time() - t_start > limit:
            timed_out = True
            stop_pending()
        time.sleep(0.05)
    reader.join(timeout=10)
    text = (chunks[0] if chunks else b"").decode()
    stderr_of = []
    for f in errs:
        f.seek(0)
        stderr_of.append(f.read().decode(errors="replace"))
        f.close()
    if failed is None and not timed_out:
        for e in stderr_of:
            sys.stderr.write(e)
        sys.stdout.write(text)
        sys.stdout.flush()
        return 0, False
    if timed_out:
        sys.stderr.write(f"bench.py: ranks still running after PLFEM_BENCH_TIMEOUT = {limit:.0f} s were stopped\n")
        return 1, False
    r, rc = failed
    rendezvous = is_rendezvous_failure(rc, text, stderr_of[r])
    how = f"was killed by signal {-rc}" if rc < 0 else f"exited with code {rc}"
    sys.stderr.write(f"bench.py: rank {r} {how}; the end of its stderr:\n" + "".join("    " + ln + "\n" for ln in stderr_of[r].splitlines()[-25:]))
    return 1, rendezvous


# ----------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only)
# ----------------------------------------------------------------------------------------------------
def cpu_baseline(geom, mesh, all_cores: bool = False):
    """Oracle (CPU port of the reference algorithm in scikit-fem's loop shape + SciPy eigsh with the reference's
    arguments) timed on the same workload: one full solve with BLAS limited to 4 threads (the reference sets
    OMP/MKL_NUM_THREADS=4, main.py:19-20).  ``--cpu-all-cores`` adds the same solve with BLAS unrestricted (SURVEY.md
    section 8d asks for both; assembly and SuperLU are single-threaded and 256 BLAS threads make eigsh slower: 0.7-0.8
    against 1.2-1.35 modes/s in rounds 2 and 3) -- off by default: it doubled the CPU share of the default run, during
    which the GPU idles.  The GPU steps run BEFORE and AFTER this leg (run_solve_config), so that the
    driver's GPU-activity samples see the device at both ends of the run.  Returns (cpu_baseline object, reference modes)."""
    from oracle import hfield
    from oracle.p2 import MeshTriLite

    def one(threads):
        limiter = None
        if threads is not None:
            try:
                from threadpoolctl import threadpool_limits
                limiter = threadpool_limits(limits=threads)
            except Exception:                                         # pragma: no cover
                limiter = None
        tm = {}
        t0 = time.perf_counter()
        ref = hfield.solve_vectorial_modes(geom, MeshTriLite(mesh.p, mesh.t), N_MODES, fused=False, timings=tm)
        dt = time.perf_counter() - t0
        if limiter is not None:
            limiter.restore_original_limits()
        return ref, dt, tm

    threads = int(os.environ.get("PLFEM_CPU_THREADS", "4"))
    ref, dt, tm = one(threads)
    base = {"value": N_MODES / dt, "unit": "modes/s", "cores": threads, "kind": "port",
            "sample": f"1 full solve of the same workload ({dt:.1f} s: assembly {tm['assembly']:.1f} s, "
                      f"eigsh {tm['eigsh']:.1f} s; assembly and SuperLU are single-threaded, BLAS limited to {threads} threads; "
                      f"host has {os.cpu_count()} logical CPUs)"}
    if all_cores:
        _ref2, dt_all, tm_all = one(None)
        base["all_cores"] = {"value": N_MODES / dt_all, "unit": "modes/s", "cores": os.cpu_count(),
                             "sample": f"the same solve with BLAS unrestricted ({dt_all:.1f} s: assembly {tm_all['assembly']:.1f} s, "
                                       f"eigsh {tm_all['eigsh']:.1f} s)"}
    return base, ref


def parity_block(gpu_modes, ref):
    import numpy as np
    from oracle.compare import mode_field_errors
    same = len(ref) == len(gpu_modes)
    dn = max(abs(a["n_eff"] - b["n_eff"]) for a, b in zip(gpu_modes, ref)) if same else float("nan")
    worst = float(np.max(mode_field_errors(gpu_modes, ref))) if same else float("nan")
    return {"max_abs_dn_eff": dn, "max_field_l2": worst, "n_modes_compared": len(ref),
            "tolerance": {"dn_eff": 5e-5, "field_l2": 1e-6}}


# ----------------------------------------------------------------------------------------------------
# rooflines
# ----------------------------------------------------------------------------------------------------
def library_kernel_names():
    """Demangled names of every kernel in the built library (the host-side launch stubs carry them), or None when no
    ``nm`` is at hand."""
    lib = os.path.join(ROOT, "pl_fem_vectoriel_amd", "libplfem_hip.so")
    for nm in ("nm", "/opt/rocm/lib/llvm/bin/llvm-nm"):
        try:
            txt = subprocess.run([nm, "-C", lib], capture_output=True, text=True, timeout=60).stdout
        except (OSError, subprocess.SubprocessError):
            continue
        names = [ln.split("__device_stub__", 1)[1] for ln in txt.splitlines() if "__device_stub__" in ln]
        if names:
            return names
    return None


def pmc_for_levels(levels: int):
    """The committed PMC summary that belongs to THIS workload and THIS code: ``profiles/rNN_pmc_families_L{levels}.json``
    of the latest round that has one (scripts/gpu_profile_round.sh writes them per rung; the round-3 file without a
    suffix was C1 = L1 only, and bench.py used it for every rung: traffic ratios of 5.9 at L = 0 and 0.23 at L = 2,
    VERDICT r3 weak #8).  A file is refused -- traffic: null, with the reason in ``pmc_check`` -- when one of its kernel
    regexes no longer matches any kernel of the built library, or a kernel it counted no longer exists: the counters
    were collected on other code.  Returns (families, mfma, path, check)."""
    import glob
    import re
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_families_L{levels}.json")), reverse=True)
    if levels == 1:
        cands += sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_families.json")), reverse=True)
    names = library_kernel_names()
    why = "no PMC pass committed for this rung"
    for path in cands:
        d = json.load(open(path))
        fams = d.get("families", {})
        stale = None
        if names is not None:
            for fam, rec in fams.items():
                if not any(re.search(rec["regex"], n) for n in names):
                    stale = f"{os.path.basename(path)}: regex '{rec['regex']}' of family '{fam}' matches no kernel of the built library"
                    break
                have = {n.split("(")[0] for n in names}
                gone = [k for k in rec.get("kernel_names", []) if k not in have]
                if gone:
                    stale = f"{os.path.basename(path)}: kernel {gone[0]} counted there is not in the built library"
                    break
        if stale is None:
            return fams, d.get("mfma", {}), os.path.relpath(path, ROOT), "ok" if names is not None else "ok (no nm: kernel names not checked)"
        why = stale
    return {}, {}, None, why


def roofline_objects(kprof, stats, info, nv, ne, levels=1):
    """``roofline`` = the dominant kernel FAMILY of the step, the forward + backward solve sweeps (47 % of the GPU time
    of a solve; one launch per tree level and direction), and ``roofline.kernels`` = its parts and the other families.
    achieved = ALGORITHMIC bytes (or flop) / time measured live with HIP events on the launch stream during the first
    timed step (plfem_profile_*); formulas in DESIGN.md section 6.  ``traffic`` = HBM bytes from separate rocprofv3 --pmc
    passes (pmc_for_levels: the committed file of this rung, checked against the built library; 2 x FETCH_SIZE +
    WRITE_SIZE as MI355X_MICROARCH.md prescribes for gfx950) per the same unit, ``traffic_ratio`` = traffic / algorithmic
    bytes; null when this rung has no valid PMC pass."""
    pmc, pmc_mfma, pmc_path, pmc_check = pmc_for_levels(levels)

    def traffic(family):
        return pmc.get(family, {}).get("hbm_bytes")

    def hbm(name, bytes_, us, n, formula, family=None):
        a = bytes_ / (us * 1e-6) / 1e9
        tr = traffic(family) if family else None
        return {"kernel": name, "bound": "hbm", "achieved": a, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": a / HBM_PEAK_GBS,
                "ranges_timed": n, "avg_us": us / n, "algorithmic_bytes": bytes_ / n, "formula": formula,
                "traffic": tr, "traffic_ratio": (tr / (bytes_ / n)) if tr else None}

    sl = kprof["slots"]
    kernels = []
    sweep_formula = "8 B x sum over fronts (s2 m - s2^2/2 + P (m + s2)), P = 4"
    for key, name, fam in (("fwd_sweep", "forward sweep (all levels: k_fwd + k_fwd_mix + k_fwd_rows)", "forward sweep (k_fwd, k_fwd_mix, k_fwd_rows)"),
                           ("bwd_sweep", "backward sweep (all levels: k_bwd + k_bwd_rows)", "backward sweep (k_bwd, k_bwd_rows)")):
        if sl[key]["ranges"] > 0:
            kernels.append(hbm(name, sl[key]["bytes"], sl[key]["total_us"], sl[key]["ranges"], sweep_formula, fam))
    if kprof["launches"] > 0:
        kernels.append(hbm("k_fwd<4> / k_fwd_mix<4> (tile-form forward levels, per launch)", kprof["bytes"], kprof["total_us"],
                           kprof["launches"], "the level's share of the sweep formula", "k_fwd / k_fwd_mix (tile-form forward levels)"))
    if sl["spmv_b"]["ranges"] > 0:
        kernels.append(hbm("k_spmv_b_block<4>", sl["spmv_b"]["bytes"], sl["spmv_b"]["total_us"], sl["spmv_b"]["ranges"],
                           "12 B x nnz + 4 B x (N + 1) + 2 x 8 B x 4 x 2N", "k_spmv_b_block"))
    nnz = info["nnz"]
    asm_bytes = 24.0 * ne + 16.0 * nv + 8.0 * 5 * nnz
    if stats.get("assemble_us", 0) > 0:
        kernels.append(hbm("assembly (k_element_matrices + k_csr_gather)", asm_bytes, stats["assemble_us"], 1,
                           "24 B x ne (t, edge dofs) + 16 B x nv (coordinates) + 8 B x 5 x nnz (Axx Axy Ayx Ayy Minv values written once)",
                           "assembly (k_element_matrices + k_csr_gather)"))
    if stats.get("factor_us", 0) > 0:
        tf = info["factor_flops"] / (stats["factor_us"] * 1e-6) / 1e12
        mf = pmc_mfma.get("factorisation (all kernels)", {})
        kernels.append({"kernel": "factorisation (block LDL^T, all kernels)", "bound": "mfma", "achieved": tf, "peak": FP64_MFMA_PEAK_TF,
                        "unit": "TFLOP/s", "frac": tf / FP64_MFMA_PEAK_TF, "ranges_timed": 1, "avg_us": stats["factor_us"],
                        "algorithmic_flop": info["factor_flops"], "formula": "sum over fronts s2 m^2 (LDL^T + Schur complement + L11^-1 + Z)",
                        "mfma_flop_counted": mf.get("mfma_flop_per_unit"), "mfma_busy_over_sq_busy": mf.get("mfma_busy_over_sq_busy"),
                        "traffic": traffic("factorisation (all kernels)")})
    roof = None
    nf, nb = sl["fwd_sweep"]["ranges"], sl["bwd_sweep"]["ranges"]
    if nf > 0 and nb > 0:
        pairs = min(nf, nb)
        bytes_pair = sl["fwd_sweep"]["bytes"] / nf + sl["bwd_sweep"]["bytes"] / nb
        us_pair = sl["fwd_sweep"]["total_us"] / nf + sl["bwd_sweep"]["total_us"] / nb
        achieved = bytes_pair / (us_pair * 1e-6) / 1e9
        tf_, tb_ = traffic("forward sweep (k_fwd, k_fwd_mix, k_fwd_rows)"), traffic("backward sweep (k_bwd, k_bwd_rows)")
        tr = (tf_ + tb_) if (tf_ and tb_) else None
        launches_pair = 2 * (int(info.get("levels", 0)) + 1)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": tr, "traffic_ratio": (tr / bytes_pair) if tr else None,
                "traffic_source": (f"{pmc_path} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes on this workload, not this run)"
                                   if tr else None),
                "pmc_check": pmc_check,
                "kernel": "forward + backward solve sweep of the shift-invert operator, P = 4 right-hand sides (k_fwd, k_fwd_mix, "
                          "k_fwd_rows, k_bwd, k_bwd_rows: one launch per tree level and direction)",
                "unit_of_work": "one sweep pair = one application of K^-1 to 4 vectors", "pairs_timed": pairs,
                "avg_pair_us": us_pair, "algorithmic_bytes_per_pair": bytes_pair, "launches_per_pair": launches_pair,
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
        if self.world > 1 and "PLFEM_HOST_THREADS" not in os.environ:
            # every rank runs its own host analysis on a spin-waiting thread pool: share the host's cores between the ranks
            os.environ["PLFEM_HOST_THREADS"] = str(max(4, min(16, (os.cpu_count() or 64) // self.world)))
        import torch
        self.torch = torch
        if not self.fake:
            if not torch.cuda.is_available():
                raise RuntimeError("bench.py needs a GPU: the eigenmode path has no CPU fallback")
            if os.environ.get("PLFEM_BENCH_SAME_DEVICE"):
                self.local_rank = 0
            torch.cuda.set_device(self.local_rank)
        self.dist = None
        # PLFEM_BENCH_FORCE_DIST=1: a process group even for ONE rank (RANK / WORLD_SIZE / MASTER_* as torchrun sets them):
        # barrier, max-reduction and the sweep's gather then run through RCCL on a one-GPU box (tests/test_gpu_rccl_one_rank.py)
        if self.world > 1 or (os.environ.get("PLFEM_BENCH_FORCE_DIST") and "RANK" in os.environ):
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


def phase_record(ls, step_ms):
    """Where one cold step went, in ms; the entries (without ``sum`` / ``step``) add up to the step's wall time.
    Host phases are wall-clock intervals of the Python layer, device phases HIP-event intervals inside the one
    ``plfem_solve_modes`` call; ``call_gaps`` = that call's wall time minus its device phases (the wait for the index
    upload to land, host work of the Lanczos driver between its steps' launches, the tail of the mode copy behind the
    residual check); ``python`` = the rest of the step (solver construction, argument marshalling, the mode dicts and
    filters, release of the previous step's context and mode list)."""
    dev = {k: ls[k + "_us"] / 1e3 for k in ("upload", "assemble", "factor", "lanczos", "post", "residual")}
    call = ls["t_call"] * 1e3
    rec = {"symbolic_host": ls["t_symbolic"] * 1e3, "context": ls["t_context"] * 1e3, "pinned_alloc": ls["t_pinned"] * 1e3}
    rec.update({"assemble": dev["assemble"], "factor": dev["factor"], "lanczos": dev["lanczos"], "post": dev["post"],
                "residual_check": dev["residual"]})
    # the index upload overlaps the host between plfem_create and the call; what of it the call still waits for is in call_gaps
    rec["call_gaps"] = call - sum(rec[k] for k in ("assemble", "factor", "lanczos", "post", "residual_check"))
    rec["python"] = step_ms - (rec["symbolic_host"] + rec["context"] + rec["pinned_alloc"] + call)
    out = dict(rec)
    out["sum"] = sum(out.values())
    out["step"] = step_ms
    out["index_upload_on_stream"] = dev["upload"]
    return out


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
    base = ref_modes = None
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
    # and every phase of every step, so that "breakdown_ms" is the MEAN over the timed steps and adds up to ms_per_step
    step_ms, host_ms, phases = [], [], []
    t0 = time.perf_counter()
    for it in range(args.steps):
        ts = time.perf_counter()
        solver, modes = step(profile=(it == 0))
        step_ms.append((time.perf_counter() - ts) * 1e3)
        ls = solver.last_stats
        host_ms.append((ls["t_symbolic"] * 1e3, ls["t_context"] * 1e3, ls.get("t_workspace", 0.0) * 1e3))
        phases.append(phase_record(ls, step_ms[-1]))
    D.sync()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    stats = dict(solver.last_stats)
    # the CPU leg sits BETWEEN the timed GPU steps and the GPU extras below (warm steps, concurrent lanes): the device is
    # busy at both ends of the run, whichever way the driver's activity samples fall
    if world == 1 and with_cpu_baseline:
        base, ref_modes = cpu_baseline(geom, mesh, all_cores=args.cpu_all_cores)

    # steady figure -- extra info, outside the timed region: the SAME cold step repeated for a few seconds (hundreds of
    # steps instead of K): what the K timed steps are a sample of, with its spread (host jitter shows up in p95 / max)
    steady = None
    if world == 1 and levels == 1 and with_cpu_baseline and args.steady_seconds > 0:
        ms, sym_ms = [], []
        t_end = time.perf_counter() + args.steady_seconds
        while time.perf_counter() < t_end:
            ts = time.perf_counter()
            sv, _ = step()
            ms.append((time.perf_counter() - ts) * 1e3)
            sym_ms.append(sv.last_stats["t_symbolic"] * 1e3)
        total = sum(ms)
        ms.sort()
        sym_ms.sort()
        pct = lambda v, q: v[min(len(v) - 1, int(q * len(v)))]
        steady = {"steps": len(ms), "seconds": args.steady_seconds, "ms_per_step_mean": total / len(ms),
                  "ms_per_step_median": pct(ms, 0.5), "ms_per_step_p95": pct(ms, 0.95),
                  "ms_per_step_min": ms[0], "ms_per_step_max": ms[-1], "modes_per_s": N_MODES * len(ms) / (total * 1e-3),
                  # the host analysis inside those steps: the part of a cold step that depends on the (shared) host
                  "symbolic_host_ms": {"median": pct(sym_ms, 0.5), "p95": pct(sym_ms, 0.95), "max": sym_ms[-1]}}

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
    roof = roofline_objects(kprof, prof_stats or stats, info, mesh.p.shape[1], mesh.t.shape[1], levels) if kprof["slots"] else None
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
        "breakdown_ms": dict({k: sum(ph[k] for ph in phases) / len(phases) for k in phases[0]}, warm_step=warm_ms),
        "lanczos": {"n_opinv": stats["n_opinv"], "restarts": stats["restarts"], "nconv": stats["nconv"],
                    "true_residual": stats.get("true_residual"), "refined": stats.get("refined")},
        "raw_eigenpairs_per_s": world * args.steps * stats["n_req"] / elapsed,
        "step_ms": [round(v, 2) for v in step_ms],
        "host_ms_max": {"symbolic": round(max(h[0] for h in host_ms), 2), "context": round(max(h[1] for h in host_ms), 2),
                        "workspace_alloc": round(max(h[2] for h in host_ms), 2)},
        "roofline": roof,
    }
    if steady is not None:
        out["steady"] = steady
    if conc is not None:
        out["concurrent"] = {"in_flight": in_flight, "value": conc, "unit": "modes/s",
                             "note": "same cold solves, several in flight on the one GPU (host thread + stream each); "
                                     "not the headline: the reference runs one solve at a time"}
    if base is not None:
        out["cpu_baseline"] = base
        out["parity"] = parity_block(modes, ref_modes)
        # (vs_baseline stays null: BASELINE.md holds no published number for this metric; this is the ratio to the CPU
        # port timed on this host, a reported baseline, not a target)
        out["vs_cpu_baseline"] = {"cores_4": out["value"] / base["value"]}
        if "all_cores" in base:
            out["vs_cpu_baseline"]["all_cores"] = out["value"] / base["all_cores"]["value"]
    return out


def _host_picture(solve, wall):
    """Host-side picture of rank 0's lanes over the timed sweeps: share of the wall time a lane spends inside a solve call,
    of that the wait for a mesh's analysis (prepared ahead by background threads), the number of contexts made, and what
    the preparers spent per mesh on the mesh itself and on its analysis."""
    tl = list(solve.timeline)
    if not tl:
        return None
    lanes_seen = sorted({t[0] for t in tl})        # lane INDICES (0 .. lanes - 1), the same over all steps of the run
    host = {"lanes": len(lanes_seen),
            "busy_fraction_per_lane": [round(sum(t[4] - t[2] for t in tl if t[0] == ln) / wall, 3) for ln in lanes_seen],
            "analysis_wait_ms_total": round(1e3 * sum(t[3] - t[2] for t in tl), 2),
            "contexts_created": int(sum(1 for t in tl if t[5])), "solves": len(tl)}
    pre = list(getattr(solve, "prepared", []))
    if pre:
        host["prepare_ms_per_mesh"] = {"mesh": round(1e3 * sum(p[2] - p[1] for p in pre) / len(pre), 2),
                                       "analysis": round(1e3 * sum(p[3] - p[2] for p in pre) / len(pre), 2), "meshes": len(pre)}
    return host


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
    n_local = 0
    for _ in range(args.warmup):
        table, n_local = run_sweep(items, D.rank, world, solve=solve, device=device, lanes=lanes)
    D.sync()
    if hasattr(solve, "timeline"):
        solve.timeline.clear()
        solve.prepared.clear()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        table, n_local = run_sweep(items, D.rank, world, solve=solve, device=device, lanes=lanes)
    D.sync()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    host = _host_picture(solve, time.perf_counter() - t0) if hasattr(solve, "timeline") else None
    # the same sweep with the mesh producer INSIDE the timed region (Delaunay + refinement per cross-section, overlapped
    # with the solves by the sweep's preparer threads): reported beside the headline, which keeps its inputs resident
    elapsed_mesh = host_mesh = None
    if not D.fake:
        solve_m = default_solve(D.local_rank, meshes=None)
        D.sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            table_m, _n = run_sweep(items, D.rank, world, solve=solve_m, device=device, lanes=lanes)
        D.sync()
        wall_m = time.perf_counter() - t1
        elapsed_mesh, host_mesh = D.max_over_ranks(wall_m), _host_picture(solve_m, wall_m)
        if D.rank == 0:
            assert all(np.array_equal(table_m[i], table[i]) for i in table)
    if D.rank != 0:
        return None
    assert sorted(table) == list(range(len(items)))
    out = {"metric": METRIC, "value": args.steps * len(items) * N_MODES / elapsed, "unit": "modes/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "fake (rehearsal)" if D.fake else "synthetic",
           "config": {"workload": "C4: 64-solve multi-band sweep = 16 cross-sections (12 multi-core layouts at pitch 8 um + 7-core at "
                                  "pitch 6/7/9/10 um) x lambda in {1490, 1550, 1600, 1650} nm, 10 modes each, meshes as C1",
                      "step": "one whole sweep: per mesh symbolic once (shared by the lanes), per lane and mesh a context, per "
                              "wavelength assembly + factor + Lanczos + check + post; one all-gather of 64 fixed-size records",
                      "parallelism": f"{world} rank(s), {len(items) // world} solves per GPU, {lanes} in flight per GPU, "
                                     f"backend {D.backend}",
                      "host_threads_per_rank": int(os.environ.get("PLFEM_HOST_THREADS", "0")) or None},
           "sweep": {"solves": len(items), "solves_per_s": args.steps * len(items) / elapsed, "solves_rank0": n_local,
                     "n_eff_checksum": float(sum(float(np.sum(table[i])) for i in sorted(table)))}}
    if host is not None:
        out["sweep"]["host_timeline_rank0"] = host
    if elapsed_mesh is not None:
        out["sweep"]["with_mesh_production"] = {"ms_per_step": elapsed_mesh / args.steps * 1e3,
                                                "value": args.steps * len(items) * N_MODES / elapsed_mesh, "unit": "modes/s",
                                                "host_timeline_rank0": host_mesh,
                                                "note": "MeshGenerator recipe (Delaunay + 1 refinement) of the 16 cross-sections inside the "
                                                        "timed region, prepared ahead of the lanes by up to 4 host threads"}
    return out


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
