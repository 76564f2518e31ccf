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

K_MAX = 48          # padded record width: up to 48 effective indices per solve
REC_WIDTH = K_MAX + 4
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
    """Static partition: groups sharing a mesh stay together; groups dealt heaviest-first to the least
    loaded rank that still has room (longest-processing-time rule under an equal-count cap).
    Deterministic — every rank computes the same table."""
    groups: Dict[tuple, List[SweepItem]] = {}
    for it in items:
        groups.setdefault(it.mesh_key, []).append(it)
    order = sorted(groups.values(), key=lambda g: (-sum(i.cost() for i in g), g[0].index))
    cap = -(-len(order) // world_size)          # at most ceil(G / world_size) meshes per rank (8 solves per GPU in C4)
    loads = [0.0] * world_size
    counts = [0] * world_size
    out: List[List[SweepItem]] = [[] for _ in range(world_size)]
    for g in order:
        r = min((q for q in range(world_size) if counts[q] < cap), key=lambda q: (loads[q], q))
        out[r].extend(g)
        loads[r] += sum(i.cost() for i in g)
        counts[r] += 1
    return out


def default_solve(device: Optional[int] = None, meshes: Optional[dict] = None) -> Callable[[SweepItem, dict], np.ndarray]:
    """Solve one item on the GPU; ``cache`` keeps one solver (symbolic analysis + context) per mesh.
    ``meshes``: optional ``{item.mesh_key: TriMesh}`` of meshes produced beforehand (bench.py keeps the mesh
    producer, the step before the path, outside its timed region); missing keys are generated on demand.

    ``solve.prepare(item, cache)`` may be called from a background thread for the NEXT mesh: mesh
    generation (SciPy Delaunay + native refinement) and the host-side symbolic analysis release the GIL,
    so they overlap with the GPU solves of the current mesh."""
    from . import _native
    from .mesh import generate_mesh
    from .solver_fem import TrueVectorialMaxwellSolver

    def build(item: SweepItem) -> dict:
        g = item.geometry()
        mesh = (meshes or {}).get(item.mesh_key)
        if mesh is None:
            mesh = generate_mesh(g, item.mesh_refinement, item.mesh_levels)
        return {"key": item.mesh_key, "mesh": mesh, "sym": _native.Symbolic(mesh.p, mesh.t)}

    def prepare(item: SweepItem, cache: dict) -> None:
        cache.setdefault("prefetched", {})[item.mesh_key] = build(item)

    def solve(item: SweepItem, cache: dict) -> np.ndarray:
        g = item.geometry()
        ent = cache.get("cur")
        if ent is None or ent["key"] != item.mesh_key:
            if ent is not None:
                ent["solver"].clear_cache()
            ent = cache.get("prefetched", {}).pop(item.mesh_key, None) or build(item)
            ent["solver"] = TrueVectorialMaxwellSolver(g, device=device)
            ent["solver"].adopt_analysis(ent["mesh"], ent["sym"])
            cache["cur"] = ent
        s = ent["solver"]
        s.geometry, s.k0 = g, g.k0
        modes = s.solve_vectorial_modes(ent["mesh"], item.n_modes)
        return np.array([m["n_eff"] for m in modes], dtype=np.float64)

    solve.prepare = prepare
    return solve


def _run_lane(entries, solve, rank: int, rec: np.ndarray, device, own_stream: bool, errors: list) -> None:
    """One lane = a sequential pass over ``entries`` [(row in rec, item), ...] with its own solver cache (and, on
    a GPU, its own stream so that lanes overlap on the device); the next different mesh is prepared on a host
    thread while the current one is being solved."""
    import contextlib
    import threading

    ctx = contextlib.nullcontext()
    if own_stream:
        import torch
        ctx = torch.cuda.stream(torch.cuda.Stream(device))
    cache: dict = {}
    prefetch = None
    with ctx:
        for pos, (q, it) in enumerate(entries):
            new_group = pos == 0 or entries[pos - 1][1].mesh_key != it.mesh_key
            if new_group:
                if prefetch is not None:
                    prefetch.join()             # the mesh prepared in the background is this item's
                    prefetch = None
                if hasattr(solve, "prepare"):
                    # while the GPU works on this mesh, prepare the next DIFFERENT mesh on a host thread
                    nxt = next((x for _, x in entries[pos + 1:] if x.mesh_key != it.mesh_key), None)
                    if nxt is not None:
                        prefetch = threading.Thread(target=solve.prepare, args=(nxt, cache), daemon=True)
                        prefetch.start()
            # an exception in one solve must not keep this rank from the collective (the other ranks would block in it
            # forever): it becomes a status code in the item's record, n_eff stays NaN, and run_sweep raises on every
            # rank after the gather
            rec[q, 0], rec[q, 1], rec[q, 2] = it.index, 0, rank
            try:
                ne = np.asarray(solve(it, cache), dtype=np.float64)[:K_MAX]
                rec[q, 1], rec[q, 3] = len(ne), ST_OK
                rec[q, 4:4 + len(ne)] = ne
            except Exception as exc:                       # noqa: BLE001 - reported through the record
                from scipy.sparse.linalg import ArpackNoConvergence
                rec[q, 3] = ST_NOCONV if isinstance(exc, ArpackNoConvergence) else ST_ERROR
                errors.append((it.index, exc))
                ent = cache.pop("cur", None)               # the context may be in an undefined state: drop it
                if ent is not None and "solver" in ent:
                    try:
                        ent["solver"].clear_cache()
                    except Exception:                      # noqa: BLE001
                        pass
        if prefetch is not None:
            prefetch.join()
        ent = cache.get("cur")
        if ent is not None and "solver" in ent:
            ent["solver"].clear_cache()


def run_sweep(items: Sequence[SweepItem], rank: int = 0, world_size: int = 1,
              solve: Optional[Callable[[SweepItem, dict], np.ndarray]] = None, device=None,
              gather: bool = True, lanes: int = 1):
    """Solve this rank's share and gather fixed-size records on every rank.

    Returns ``(table, local_count)``; ``table[i]`` is the descending n_eff list of item ``i``
    (available on all ranks after the gather).  Records are ``[index, count, rank, 0, n_eff...]``
    padded with NaN to ``REC_WIDTH`` doubles — the only inter-GPU traffic of the whole sweep.

    ``lanes`` > 1: that many solves in flight on this rank's GPU, each lane a host thread with its own context and
    stream (whole cross-sections are dealt to the lanes, so a lane still reuses its analysis across wavelengths).
    A 1e5-DOF solve leaves most of an MI355X idle (host analysis, latency-bound tree levels); two lanes fill it.
    """
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
    if lanes == 1:
        try:
            _run_lane(list(enumerate(mine)), solve, rank, rec, device, own_stream=False, errors=errors)
        except Exception as exc:               # noqa: BLE001 - e.g. mesh preparation failed: still reach the gather
            errors.append((-1, exc))
    else:
        import threading
        # deal whole mesh groups to the lanes, heaviest first, always to the lane with the least work so far
        groups: Dict[tuple, list] = {}
        for q, it in enumerate(mine):
            groups.setdefault(it.mesh_key, []).append((q, it))
        order = sorted(groups.values(), key=lambda g: -sum(it.cost() for _, it in g))
        lane_items = [[] for _ in range(lanes)]
        load = [0.0] * lanes
        for g in order:
            k = int(np.argmin(load))
            lane_items[k].extend(g)
            load[k] += sum(it.cost() for _, it in g)
        on_gpu = device is not None and torch.cuda.is_available()

        def work(entries):
            try:
                _run_lane(entries, solve, rank, rec, device, own_stream=on_gpu, errors=errors)
            except Exception as exc:           # noqa: BLE001 - the lane's remaining items stay ST_SKIPPED
                errors.append((-1, exc))

        threads = [threading.Thread(target=work, args=(e,)) for e in lane_items if e]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    table: Dict[int, np.ndarray] = {}
    if world_size > 1 and gather:
        import torch.distributed as dist

        dev = torch.device("cuda", device) if (device is not None and dist.get_backend() == "nccl") else torch.device("cpu")
        local = torch.from_numpy(rec).to(dev)
        bufs = [torch.empty_like(local) for _ in range(world_size)]
        dist.all_gather(bufs, local)
        allrec = torch.stack(bufs).cpu().numpy().reshape(-1, REC_WIDTH)
    else:
        allrec = rec
    failures = []
    for row in allrec:
        if row[0] >= 0:
            if int(row[3]) == ST_OK:
                table[int(row[0])] = row[4:4 + int(row[1])].copy()
            else:
                failures.append((int(row[0]), int(row[2]), int(row[3])))
    if failures:
        err = SweepError(sorted(failures), table)
        if errors:
            raise err from errors[0][1]
        raise err
    return table, len(mine)
