"""Parametric sweep of independent cross-sections over the GPUs of one node.

The reference has no distributed code: every cross-section (arrangement, pitch,
wavelength) is an independent call of ``solve_vectorial_modes`` (``solver_fem.py:171`` keeps no state
between calls).  The path therefore shards across *solves* only: one process per GPU
(``torch.distributed``, backend ``nccl`` = RCCL over xGMI), a static cost-balanced partition, no
collective on the data path, and one all-gather of fixed-size padded records at the end
(SURVEY.md §8e).  The four wavelengths of one mesh stay on one rank so the mesh-only analysis, the
device context and its HBM workspaces are built once per mesh.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from .geometry import ARRANGEMENTS, MCFGeometry

K_MAX = 48          # padded record width: up to 48 modes per solve
# per-mode columns of a record: what LossCalculator.calculate_physical_losses reads from a mode dict
# (reference losses.py:742-825 -> n_eff, beta = n_eff k0, P_x, P_y, PDL_dB, confinement) + the filter quantity div_ratio
FIELDS = ("n_eff", "P_x", "P_y", "PDL_dB", "confinement", "div_ratio")
NF = len(FIELDS)
REC_WIDTH = 4 + NF * K_MAX
# record[3]: status of the solve -- every rank always reaches the gather, errors travel in the record
ST_OK, ST_NOCONV, ST_ERROR, ST_SKIPPED = 0, 1, 2, 3


class SweepError(RuntimeError):
    """Raised on EVERY rank after the gather when any record of the sweep carries an error status;
    ``failures`` = [(item index, rank, status), ...], ``table`` = the records that did succeed."""

    def __init__(self, failures, table):
        names = {ST_NOCONV: "no convergence", ST_ERROR: "error", ST_SKIPPED: "skipped after an earlier error"}
        super().__init__("sweep: " + ", ".join(f"item {i} on rank {r}: {names.get(s, s)}" for i, r, s in failures[:8])
                         + (" ..." if len(failures) > 8 else ""))
        self.failures = failures
        self.table = table


@dataclass(frozen=True)
class SweepItem:
    index: int
    arrangement: str
    pitch_um: float
    wavelength_um: float
    core_radius_um: float = 1.5
    n_core: float = 1.535
    n_clad: float = 1.0
    n_modes: int = 10
    mesh_refinement: float = 1.0
    mesh_levels: int = 1

    @property
    def mesh_key(self):
        return (self.arrangement, self.pitch_um, self.core_radius_um, self.mesh_refinement, self.mesh_levels)

    def geometry(self) -> MCFGeometry:
        n, variant = ARRANGEMENTS[self.arrangement]
        return MCFGeometry(n, self.pitch_um, self.core_radius_um, self.n_core, self.n_clad,
                           wavelength_um=self.wavelength_um, variant=variant)

    def cost(self) -> float:
        """~ N^1.5 with N ~ number of mesh points (cores dominate the point recipe)."""
        n = ARRANGEMENTS[self.arrangement][0]
        pts = 2000.0 + 512.0 * n
        return (pts * 4 ** self.mesh_levels) ** 1.5


def multiband_sweep_items(n_modes: int = 10, mesh_levels: int = 1, mesh_refinement: float = 1.0) -> List[SweepItem]:
    """BASELINE.json config 4: 16 cross-sections x 4 wavelengths = 64 independent solves
    (the 12 multi-core layouts at pitch 8 um + 4 pitch variants of the 7-core layout)."""
    sections = [(a, 8.0) for a in ARRANGEMENTS if a != "single_1"]
    sections += [("hexagonal_1plus6_7", p) for p in (6.0, 7.0, 9.0, 10.0)]
    items = []
    for arr, pitch in sections:
        for lam in (1.49, 1.55, 1.60, 1.65):
            items.append(SweepItem(len(items), arr, pitch, lam, n_modes=n_modes, mesh_levels=mesh_levels,
                                   mesh_refinement=mesh_refinement))
    return items


def partition(items: Sequence[SweepItem], world_size: int) -> List[List[SweepItem]]:
    """Static partition with equal item counts per rank (C4: 8 solves per GPU).  Items sharing a mesh form a group and
    stay together on one rank (the host analysis is built once per mesh and rank) -- except that a group heavier than
    three quarters of a rank's fair share is split in two halves first: the 19-core cross-section alone is 1.34 fair
    shares at 8 ranks, so whole groups leave one rank with 1.5 x the mean load and the sweep waits for it, while two
    wavelengths of it per rank cost one more analysis (10-20 ms of host time) and bring max / mean to 1.14.  The units are
    dealt heaviest-first to the least loaded rank with room (longest-processing-time rule under the count cap; a unit
    that fits nowhere is halved again).  Deterministic -- every rank computes the same table."""
    groups: Dict[tuple, List[SweepItem]] = {}
    for it in items:
        groups.setdefault(it.mesh_key, []).append(it)
    share = sum(i.cost() for i in items) / max(world_size, 1)
    cap = -(-len(items) // world_size)          # items per rank
    units: List[List[SweepItem]] = []
    for g in groups.values():
        if world_size > 1 and len(g) >= 2 and sum(i.cost() for i in g) > 0.75 * share:
            units += [g[:len(g) // 2], g[len(g) // 2:]]
        else:
            units.append(g)
    pending = sorted(units, key=lambda g: (-sum(i.cost() for i in g), g[0].index))
    loads = [0.0] * world_size
    counts = [0] * world_size
    out: List[List[SweepItem]] = [[] for _ in range(world_size)]
    while pending:
        g = pending.pop(0)
        rooms = [q for q in range(world_size) if counts[q] + len(g) <= cap]
        if not rooms:                           # (only units of several items can be stranded: halve and retry)
            pending = [g[:len(g) // 2], g[len(g) // 2:]] + pending
            continue
        r = min(rooms, key=lambda q: (loads[q], q))
        out[r].extend(g)
        loads[r] += sum(i.cost() for i in g)
        counts[r] += len(g)
    return out


def default_solve(device: Optional[int] = None, meshes: Optional[dict] = None) -> Callable[[SweepItem, dict], np.ndarray]:
    """Solve one item on the GPU.  Returns the (NF, k) array of per-mode columns ``FIELDS`` (n_eff descending).

    ``cache`` is the calling LANE's dictionary: it keeps one solver (device context) for the lane's current mesh.  The
    mesh itself and its host analysis (``Symbolic``: numbering, pattern, front tree) are built ONCE per mesh, whichever
    lane or background thread asks first (``solve.prepare(item)``), and shared by every lane: the four wavelengths of a
    cross-section can then run on four lanes, each with its own context on the one analysis.  ``meshes``: optional
    ``{item.mesh_key: TriMesh}`` of meshes produced beforehand (bench.py keeps the mesh producer, the step before the
    path, outside its timed region); missing keys are generated on demand.  Mesh generation (Qhull + native refinement)
    and the analysis release the GIL, so a background ``prepare`` overlaps with the GPU solves: with the mesh producer
    inside the timed region the 64-solve sweep takes 0.03-0.06 s longer (bench.py --sweep, ``with_mesh_production``; worker
    processes for the Delaunay stage were tried and measured the same as the threads)."""
    import threading

    from . import _native
    from .mesh import generate_mesh
    from .solver_fem import TrueVectorialMaxwellSolver

    import time as _time

    lock = threading.Lock()
    shared: Dict[tuple, dict] = {}          # mesh_key -> {"ready": Event, "mesh", "sym", "error"}
    import collections
    # host-side picture of the last few thousand solves (bounded: a long-lived solve closure must not grow without limit)
    timeline = collections.deque(maxlen=4096)   # (lane index, item index, t_start, t_analysis_ready, t_end, new context?)
    prepared = collections.deque(maxlen=1024)   # (mesh_key, t_start, t_mesh_ready, t_analysis_ready) of every mesh made here

    def prepare(item: SweepItem) -> dict:
        with lock:
            ent = shared.get(item.mesh_key)
            mine = ent is None
            if mine:
                ent = shared[item.mesh_key] = {"ready": threading.Event(), "error": None}
        if mine:
            try:
                t0 = _time.perf_counter()
                mesh = (meshes or {}).get(item.mesh_key)
                if mesh is None:
                    mesh = generate_mesh(item.geometry(), item.mesh_refinement, item.mesh_levels)
                t1 = _time.perf_counter()
                ent["mesh"], ent["sym"] = mesh, _native.Symbolic(mesh.p, mesh.t)
                prepared.append((item.mesh_key, t0, t1, _time.perf_counter()))
            except Exception as exc:            # noqa: BLE001 - every waiter sees it
                ent["error"] = exc
            ent["ready"].set()
        ent["ready"].wait()
        if ent["error"] is not None:
            raise ent["error"]
        return ent

    def release(mesh_key) -> None:
        """The sweep driver calls this when the last item of a mesh is done: drop the shared mesh + analysis."""
        with lock:
            shared.pop(mesh_key, None)

    def solve(item: SweepItem, cache: dict) -> np.ndarray:
        t_start = t_ready = _time.perf_counter()
        g = item.geometry()
        cur = cache.get("cur")
        fresh = cur is None or cur["key"] != item.mesh_key
        if fresh:
            if cur is not None:
                cur["solver"].clear_cache()
                cache.pop("cur")
            ent = prepare(item)
            t_ready = _time.perf_counter()
            solver = TrueVectorialMaxwellSolver(g, device=device)
            solver.adopt_analysis(ent["mesh"], ent["sym"])
            cur = cache["cur"] = {"key": item.mesh_key, "mesh": ent["mesh"], "solver": solver}
        s = cur["solver"]
        s.geometry, s.k0 = g, g.k0
        modes = s.solve_vectorial_modes(cur["mesh"], item.n_modes)
        timeline.append((cache.get("lane", 0), item.index, t_start, t_ready, _time.perf_counter(), fresh))
        return np.array([[m[f] for m in modes] for f in FIELDS], dtype=np.float64).reshape(NF, len(modes))

    solve.prepare = prepare
    solve.release = release
    solve.timeline = timeline               # host-side picture of a sweep (bench.py --sweep reports lane utilisation)
    solve.prepared = prepared
    return solve


def _close_lane(cache: dict) -> None:
    cur = cache.pop("cur", None)
    if cur is not None and "solver" in cur:
        try:
            cur["solver"].clear_cache()
        except Exception:                      # noqa: BLE001
            pass


def _run_lane(queue, qlock, remaining, holders, solve, rank: int, rec: np.ndarray, device, own_stream: bool, errors: list,
              lane: int = 0) -> None:
    """One lane = a host thread (with its own stream on a GPU, so that lanes overlap on the device) that takes the next
    entry (row in rec, item) off the rank's queue until it is empty.  A lane stays on the mesh it holds a device context
    for while that mesh has items left; then it takes the first mesh NO other lane is working on (a context per lane and
    mesh costs a millisecond and gigabytes: 16 meshes on 4 lanes should make 16 contexts, not 64), and only when every
    remaining mesh is taken does it join another lane's mesh (a rank with two meshes still keeps four lanes busy)."""
    import contextlib

    ctx = contextlib.nullcontext()
    if own_stream:
        import torch
        ctx = torch.cuda.stream(torch.cuda.Stream(device))
    cache: dict = {"lane": lane}               # (the lane's index: what solve.timeline keys its entries by)
    held = None
    with ctx:
        while True:
            with qlock:
                if not queue:
                    if held is not None:
                        holders[held] -= 1
                    break
                pick = next((n for n, (_q, x) in enumerate(queue) if x.mesh_key == held), None)
                if pick is None:
                    pick = next((n for n, (_q, x) in enumerate(queue) if holders.get(x.mesh_key, 0) == 0), None)
                if pick is None:                           # every remaining mesh has a lane: join the one with most items left
                    left: Dict[tuple, int] = {}
                    for _q, x in queue:
                        left[x.mesh_key] = left.get(x.mesh_key, 0) + 1
                    best = max(left, key=lambda k: (left[k] / (1 + holders.get(k, 0)), -min(n for n, (_q, x) in enumerate(queue) if x.mesh_key == k)))
                    pick = next(n for n, (_q, x) in enumerate(queue) if x.mesh_key == best)
                q, it = queue.pop(pick)
                if it.mesh_key != held:
                    if held is not None:
                        holders[held] -= 1
                    held = it.mesh_key
                    holders[held] = holders.get(held, 0) + 1
            # an exception in one solve must not keep this rank from the collective (the other ranks would block in it
            # forever): it becomes a status code in the item's record, the columns stay NaN, and run_sweep raises on
            # every rank after the gather
            rec[q, 0], rec[q, 1], rec[q, 2] = it.index, 0, rank
            try:
                res = np.asarray(solve(it, cache), dtype=np.float64)
                if res.ndim == 1:                          # a solver that only reports n_eff
                    res = np.vstack([res[None, :], np.full((NF - 1, len(res)), np.nan)])
                res = res[:, :K_MAX]
                k = res.shape[1]
                rec[q, 1], rec[q, 3] = k, ST_OK
                for f in range(NF):
                    rec[q, 4 + f * K_MAX:4 + f * K_MAX + k] = res[f]
            except Exception as exc:                       # noqa: BLE001 - reported through the record
                from scipy.sparse.linalg import ArpackNoConvergence
                rec[q, 3] = ST_NOCONV if isinstance(exc, ArpackNoConvergence) else ST_ERROR
                errors.append((it.index, exc))
                _close_lane(cache)                         # the context may be in an undefined state: drop it
            with qlock:
                remaining[it.mesh_key] -= 1
                done = remaining[it.mesh_key] == 0
            if done and hasattr(solve, "release"):
                solve.release(it.mesh_key)
        _close_lane(cache)


def records_to_modes(row: np.ndarray, k0: float) -> List[Dict]:
    """Mode records of one gathered row, as far as the loss consumer reads them (``losses.py``: n_eff, beta, P_x, P_y,
    PDL_dB, confinement; is_vectorial) -- the vectors themselves never leave the GPU that computed them."""
    k = int(row[1])
    col = {f: row[4 + n * K_MAX:4 + n * K_MAX + k] for n, f in enumerate(FIELDS)}
    return [{"n_eff": float(col["n_eff"][j]), "beta": float(col["n_eff"][j] * k0), "P_x": float(col["P_x"][j]),
             "P_y": float(col["P_y"][j]), "PDL_dB": float(col["PDL_dB"][j]), "confinement": float(col["confinement"][j]),
             "core_overlap": float(col["confinement"][j]), "div_ratio": float(col["div_ratio"][j]), "is_vectorial": True,
             "method": "H-field_V18.10"} for j in range(k)]


def run_sweep(items: Sequence[SweepItem], rank: int = 0, world_size: int = 1,
              solve: Optional[Callable[[SweepItem, dict], np.ndarray]] = None, device=None,
              gather: bool = True, lanes: int = 1, losses: Optional[str] = None):
    """Solve this rank's share and gather fixed-size records on every rank.

    Returns ``(table, local_count)``; ``table[i]`` is the descending n_eff list of item ``i`` (available on all ranks
    after the gather).  With ``losses="mux"`` / ``"demux"`` a third value follows: ``{i: LossCalculator.
    calculate_physical_losses(modes_i, geometry_i, direction, wavelength)}`` -- the IL / MDL / PDL / crosstalk columns of
    the reference's loss consumer (``losses.py:742-825``), computed on every rank from the gathered
    records.  A record is ``[index, count, rank, status, NF x K_MAX per-mode columns (FIELDS)]`` padded with NaN to
    ``REC_WIDTH`` doubles -- the only inter-GPU traffic of the whole sweep.

    ``lanes`` > 1: that many solves in flight on this rank's GPU, each lane a host thread with its own stream and device
    context.  All lanes draw from one queue (mesh groups heaviest first, the wavelengths of a mesh adjacent) and share
    the host analysis of a mesh, so a rank that owns two cross-sections still keeps four lanes busy; one background
    thread walks ahead of them and prepares the meshes + analyses (``solve.prepare``).  A 1e5-DOF solve is a chain of
    latency-bound launches and leaves most of an MI355X idle; a few lanes fill it.
    """
    import threading

    import torch

    mine = partition(items, world_size)[rank]
    if solve is None:
        solve = default_solve(device)
    per_rank = max(len(p) for p in partition(items, world_size))
    rec = np.full((per_rank, REC_WIDTH), np.nan)
    rec[:, 0] = -1
    lanes = max(1, int(lanes))
    errors: list = []            # (item index, exception) of this rank's failed solves, for the message only
    for q, it in enumerate(mine):            # a lane that dies outside a solve leaves these as they are
        rec[q, 0], rec[q, 1], rec[q, 2], rec[q, 3] = it.index, 0, rank, ST_SKIPPED
    # queue: mesh groups heaviest first, the items of a group adjacent (in item order)
    groups: Dict[tuple, list] = {}
    for q, it in enumerate(mine):
        groups.setdefault(it.mesh_key, []).append((q, it))
    order = sorted(groups.values(), key=lambda g: (-sum(it.cost() for _, it in g), g[0][1].index))
    queue = [e for g in order for e in g]
    remaining = {k: len(g) for k, g in groups.items()}
    qlock = threading.Lock()
    on_gpu = device is not None and torch.cuda.is_available()

    holders: Dict[tuple, int] = {}
    todo = [g[0][1] for g in order]              # one item per mesh, in queue order

    def preparer():
        # walk ahead of the lanes: the next meshes + analyses are ready when a lane gets to them (Delaunay, the native
        # refinement and the analysis release the GIL; a mesh takes 50-120 ms of host time, four solves on it 60-150 ms
        # of GPU time, so one preparer cannot keep four lanes fed)
        while True:
            with qlock:
                if not todo:
                    return
                it0 = todo.pop(0)
                if remaining.get(it0.mesh_key, 0) == 0:
                    continue
            try:
                solve.prepare(it0)
            except Exception:                  # noqa: BLE001 - the lane that needs it raises the same error
                pass
            # The check above and prepare() are not one atomic step: if the lanes finished this mesh (and released it)
            # in between -- or a lane died and its items stay ST_SKIPPED -- prepare() has just rebuilt a mesh and an
            # analysis nobody will ask for or release.  Drop them again.
            with qlock:
                orphan = remaining.get(it0.mesh_key, 0) == 0
            if orphan and hasattr(solve, "release"):
                solve.release(it0.mesh_key)

    def work(lane: int = 0):
        try:
            _run_lane(queue, qlock, remaining, holders, solve, rank, rec, device, own_stream=on_gpu and lanes > 1, errors=errors,
                      lane=lane)
        except Exception as exc:               # noqa: BLE001 - the lane's remaining items stay ST_SKIPPED
            errors.append((-1, exc))

    pres = []
    if hasattr(solve, "prepare") and len(order) > 0:
        pres = [threading.Thread(target=preparer, daemon=True) for _ in range(min(lanes, len(order), 4))]
        for t in pres:
            t.start()
    if lanes == 1:
        work()
    else:
        threads = [threading.Thread(target=work, args=(ln,)) for ln in range(min(lanes, max(len(queue), 1)))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    for t in pres:
        t.join()
    if hasattr(solve, "release"):              # whatever a dead lane or a late preparer left behind: nothing outlives the sweep
        for key in groups:
            solve.release(key)
    table: Dict[int, np.ndarray] = {}
    import torch.distributed as _dist
    # (a process group of ONE rank still gathers through it: the collective path can then be exercised on a one-GPU box)
    if gather and (world_size > 1 or (_dist.is_available() and _dist.is_initialized() and _dist.get_world_size() == 1)):
        import torch.distributed as dist

        dev = torch.device("cuda", device) if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
        local = torch.from_numpy(rec).to(dev)
        bufs = [torch.empty_like(local) for _ in range(world_size)]
        dist.all_gather(bufs, local)
        allrec = torch.stack(bufs).cpu().numpy().reshape(-1, REC_WIDTH)
    else:
        allrec = rec
    failures = []
    rows: Dict[int, np.ndarray] = {}
    for row in allrec:
        if row[0] >= 0:
            if int(row[3]) == ST_OK:
                table[int(row[0])] = row[4:4 + int(row[1])].copy()
                rows[int(row[0])] = row
            else:
                failures.append((int(row[0]), int(row[2]), int(row[3])))
    if failures:
        err = SweepError(sorted(failures), table)
        if errors:
            raise err from errors[0][1]
        raise err
    if losses is None:
        return table, len(mine)
    from .losses import LossCalculator
    by_index = {it.index: it for it in items}
    loss_table = {}
    for i, row in rows.items():
        it = by_index[i]
        g = it.geometry()
        loss_table[i] = LossCalculator.calculate_physical_losses(records_to_modes(row, g.k0), g, losses, 1e3 * it.wavelength_um)
    return table, len(mine), loss_table
