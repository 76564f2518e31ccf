"""Multi-process path of the parametric sweep on CPU: world_size 2, gloo backend, fake solver
(the sharding, the static partition and the gather are what is under test, not the eigen-solve)."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp

import pytest

from pl_fem_vectoriel_amd.sweep import (FIELDS, K_MAX, NF, ST_ERROR, ST_NOCONV, SweepError, SweepItem, multiband_sweep_items,
                                        partition, run_sweep)


def fake_solve(item: SweepItem, cache: dict) -> np.ndarray:
    cache["calls"] = cache.get("calls", 0) + 1
    k = 3 + item.index % 5
    return 1.26 + 1e-3 * item.index - 1e-5 * np.arange(k) + 1e-7 * item.wavelength_um


def test_partition_covers_everything_once_and_keeps_meshes_together():
    items = multiband_sweep_items()
    assert len(items) == 64                                    # BASELINE.json config 4
    for ws in (1, 2, 4, 8):
        parts = partition(items, ws)
        flat = sorted(i.index for p in parts for i in p)
        assert flat == list(range(64))
        homes = {}
        for r, p in enumerate(parts):
            keys = [i.mesh_key for i in p]
            # the wavelengths of a mesh a rank holds are contiguous; a mesh is whole on one rank or (heavy meshes at
            # many ranks) in two halves on two
            for k in set(keys):
                idx = [q for q, kk in enumerate(keys) if kk == k]
                assert len(idx) in (2, 4) and idx == list(range(idx[0], idx[0] + len(idx)))
                homes.setdefault(k, []).append(r)
        assert all(len(v) <= 2 for v in homes.values())
        assert all(len(p) == 64 // ws for p in parts)          # equal counts: 8 solves per GPU at 8 ranks
        loads = [sum(i.cost() for i in p) for p in parts]
        # (whole groups only: the 19-core mesh alone made one rank 1.5 x the mean share at 8 ranks)
        assert max(loads) / (sum(loads) / ws) < 1.15 and max(loads) / min(loads) <= 1.3, loads


def _worker(rank, world_size, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    items = multiband_sweep_items()[:24]
    table, n_local = run_sweep(items, rank, world_size, solve=fake_solve)
    dist.barrier()
    np.save(os.path.join(out_dir, f"r{rank}.npy"),
            np.array([[i, len(table[i]), table[i][0]] for i in sorted(table)]))
    np.save(os.path.join(out_dir, f"n{rank}.npy"), np.array([n_local]))
    dist.destroy_process_group()


def test_two_rank_gloo_sweep(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    items = multiband_sweep_items()[:24]
    ref, _ = run_sweep(items, 0, 1, solve=fake_solve)
    tabs = [np.load(tmp_path / f"r{r}.npy") for r in range(2)]
    np.testing.assert_array_equal(tabs[0], tabs[1])           # every rank holds the full table after the gather
    assert len(tabs[0]) == 24
    for i, cnt, first in tabs[0]:
        assert int(cnt) == len(ref[int(i)]) and first == ref[int(i)][0]
    n = [int(np.load(tmp_path / f"n{r}.npy")[0]) for r in range(2)]
    assert sum(n) == 24 and min(n) >= 8
    assert K_MAX >= 32


def failing_solve(item: SweepItem, cache: dict) -> np.ndarray:
    from pl_fem_vectoriel_amd._native import ArpackLikeNoConvergence
    if item.index == 5:
        raise ArpackLikeNoConvergence("no convergence (test)", None, None)
    if item.index == 13:
        raise RuntimeError("HIP error (test)")
    return fake_solve(item, cache)


def _worker_failing(rank, world_size, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    items = multiband_sweep_items()[:24]
    try:
        run_sweep(items, rank, world_size, solve=failing_solve)
        outcome = "no exception"
    except SweepError as e:       # raised on BOTH ranks, after the collective
        outcome = repr(e.failures) + f" ok={len(e.table)}"
    dist.barrier()                # neither rank is stuck in the all_gather
    with open(os.path.join(out_dir, f"f{rank}.txt"), "w") as fh:
        fh.write(outcome)
    dist.destroy_process_group()


def test_failed_solve_reaches_the_gather_and_raises_on_every_rank(tmp_path):
    """an exception in one rank's solve used to skip the all_gather and hang the other ranks."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_failing, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    outs = [open(tmp_path / f"f{r}.txt").read() for r in range(2)]
    assert outs[0] == outs[1]
    parts = partition(multiband_sweep_items()[:24], 2)
    rank_of = {it.index: r for r, p in enumerate(parts) for it in p}
    assert outs[0] == repr([(5, rank_of[5], ST_NOCONV), (13, rank_of[13], ST_ERROR)]) + " ok=22"
    # single process: same behaviour, the original exception is chained
    with pytest.raises(SweepError) as ei:
        run_sweep(multiband_sweep_items()[:24], 0, 1, solve=failing_solve)
    assert [f[0] for f in ei.value.failures] == [5, 13] and ei.value.__cause__ is not None


# ---- records that feed the loss consumer -----------------------------------------
def fake_solve_full(item: SweepItem, cache: dict) -> np.ndarray:
    """(NF, k) per-mode columns like default_solve returns them: n_eff descending, P_x + P_y = 1, ..."""
    k = 6 + item.index % 7
    rng = np.random.default_rng(1000 + item.index)
    n_eff = np.sort(1.26 + 0.004 * rng.random(k))[::-1]
    px = rng.uniform(0.3, 0.7, k)
    pdl = np.clip(10 * np.log10(np.maximum(px, 1 - px) / np.minimum(px, 1 - px)), 0, 50)
    return np.stack([n_eff, px, 1 - px, pdl, rng.uniform(0.4, 0.8, k), rng.uniform(0.02, 0.04, k)])


def _direct_losses(item, direction):
    from pl_fem_vectoriel_amd.losses import LossCalculator
    g = item.geometry()
    cols = fake_solve_full(item, {})
    modes = [{"n_eff": float(cols[0, j]), "beta": float(cols[0, j] * g.k0), "P_x": float(cols[1, j]), "P_y": float(cols[2, j]),
              "PDL_dB": float(cols[3, j]), "confinement": float(cols[4, j]), "core_overlap": float(cols[4, j]),
              "div_ratio": float(cols[5, j]), "is_vectorial": True} for j in range(cols.shape[1])]
    return LossCalculator.calculate_physical_losses(modes, g, direction, 1e3 * item.wavelength_um)


def _worker_losses(rank, world_size, port, out_dir):
    import json
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    items = multiband_sweep_items()[:24]
    table, n_local, losses = run_sweep(items, rank, world_size, solve=fake_solve_full, lanes=2, losses="demux")
    dist.barrier()
    with open(os.path.join(out_dir, f"l{rank}.json"), "w") as fh:
        json.dump({str(i): losses[i] for i in sorted(losses)}, fh)
    dist.destroy_process_group()


def test_gathered_records_yield_the_loss_columns_on_every_rank(tmp_path):
    import json
    assert NF == 6 and FIELDS[0] == "n_eff"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker_losses, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = [json.load(open(tmp_path / f"l{r}.json")) for r in range(2)]
    assert got[0] == got[1] and len(got[0]) == 24            # every rank holds every item's loss columns
    items = multiband_sweep_items()[:24]
    for it in items:
        want = _direct_losses(it, "demux")
        have = got[0][str(it.index)]
        assert have["success"] and set(have) == set(want)
        for key, v in want.items():
            if isinstance(v, float):
                assert abs(have[key] - v) <= 1e-12 * max(1.0, abs(v)), (it.index, key)
            else:
                assert have[key] == v, (it.index, key)
    # single process, one lane, the other direction
    table, n, losses = run_sweep(items, 0, 1, solve=fake_solve_full, losses="mux")
    assert n == 24 and losses[7]["direction"] == "mux" and abs(losses[7]["IL_dB"] - _direct_losses(items[7], "mux")["IL_dB"]) < 1e-12


def test_four_lanes_stay_busy_on_a_rank_with_two_meshes():
    """World size 8: a rank owns two cross-sections x four wavelengths.  The lanes draw single items from one queue and
    share a mesh's analysis, so all four work (whole groups per lane left two of four idle)."""
    import threading
    import time
    items = multiband_sweep_items()
    parts = partition(items, 8)
    for mine in parts:
        assert len(mine) == 8 and len({i.mesh_key for i in mine}) in (2, 3)     # (3: halves of the heaviest meshes)
    rank = next(r for r, mine in enumerate(parts) if len({i.mesh_key for i in mine}) == 2)
    seen, prepared, lock = {}, [], threading.Lock()

    def solve(item, cache):
        time.sleep(0.03)
        with lock:
            seen.setdefault(threading.get_ident(), []).append(item.index)
        return fake_solve(item, cache)

    def prepare(item):
        with lock:
            prepared.append(item.mesh_key)
    solve.prepare = prepare
    table, n = run_sweep(items, rank, 8, solve=solve, gather=False, lanes=4)
    assert n == 8 and len(table) == 8
    assert len(seen) == 4 and sorted(len(v) for v in seen.values()) == [2, 2, 2, 2]     # four lanes, two solves each
    assert len(set(prepared)) == 2                                                       # one preparation per mesh



def test_lanes_keep_one_context_per_mesh_while_meshes_outnumber_them():
    """One rank, 16 meshes, 4 lanes: a lane takes a mesh nobody else is on while there is one, so the 64 solves need
    about 16 (lane, mesh) contexts -- not the 64 that four lanes sharing every mesh would create."""
    import threading
    import time
    items = multiband_sweep_items()
    pairs, lock = set(), threading.Lock()

    def solve(item, cache):
        time.sleep(0.002)
        with lock:
            pairs.add((threading.get_ident(), item.mesh_key))
        return fake_solve(item, cache)

    table, n = run_sweep(items, 0, 1, solve=solve, lanes=4)
    assert n == 64 and len(table) == 64
    assert len({k for _t, k in pairs}) == 16 and len({t for t, _k in pairs}) == 4
    assert len(pairs) <= 16 + 3                      # (only the tail, when fewer meshes than lanes remain, is shared)


def test_a_late_preparer_leaves_nothing_behind():
    """the preparer checks `remaining` and calls prepare() in two steps.  When the lanes finish (and release) a
    mesh in between, prepare() rebuilds a mesh + analysis nobody asks for: they must be released again, and nothing may
    outlive run_sweep in the solve closure's `shared` table.  Also: eight ranks get eight solves each (two or three
    meshes: the heaviest cross-sections are split in halves) with a cost spread within 1.3."""
    import threading
    import time
    items = multiband_sweep_items()
    shared, lock, log = set(), threading.Lock(), []

    def solve(item, cache):                     # instant: the lanes are done long before a preparer returns
        return fake_solve(item, cache)

    def prepare(item):
        time.sleep(0.05)
        with lock:
            shared.add(item.mesh_key)
            log.append(("prepare", item.mesh_key))

    def release(key):
        with lock:
            shared.discard(key)
            log.append(("release", key))
    solve.prepare, solve.release = prepare, release
    table, n = run_sweep(items, 0, 4, solve=solve, gather=False, lanes=4)
    assert n == 16 and len(table) == 16
    assert not shared, shared                   # every prepared mesh was released again
    assert any(kind == "prepare" for kind, _ in log)
    parts = partition(items, 8)
    assert all(len(p) == 8 and len({i.mesh_key for i in p}) in (2, 3) for p in parts)
    loads = [sum(i.cost() for i in p) for p in parts]
    assert max(loads) / min(loads) <= 1.3, loads
