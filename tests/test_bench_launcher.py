"""`python bench.py --gpus N` must really run N ranks (`--gpus` was parsed and ignored).  The
launcher, the barrier + max-over-ranks timing and the sweep's gather are rehearsed here without any GPU:
PLFEM_BENCH_FAKE=1 replaces the solve by a sleep, PLFEM_BENCH_BACKEND=gloo replaces RCCL."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, PLFEM_BENCH_FAKE="1", PLFEM_BENCH_BACKEND="gloo")
for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    ENV.pop(k, None)


def _run(*flags, env=ENV, timeout=300):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_2_launches_two_ranks_and_prints_one_line():
    res = _run("--gpus", "2", "--steps", "4", "--warmup", "1")
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                   # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
    # 2 ranks x 4 steps x 10 modes over ~4 x 5 ms, max over ranks
    assert abs(d["value"] - 2 * 4 * 10 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]
    assert 4.0 < d["ms_per_step"] < 200.0


def test_sweep_mode_shards_the_64_solves_over_the_ranks():
    one = _run("--gpus", "1", "--steps", "1", "--warmup", "0", "--sweep")
    two = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--sweep")
    assert one.returncode == 0 and two.returncode == 0, one.stderr[-2000:] + two.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    d2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert d1["n_gpus"] == 1 and d2["n_gpus"] == 2 and d1["scaling"] == d2["scaling"] == "strong"
    assert d1["sweep"]["solves"] == d2["sweep"]["solves"] == 64
    assert d1["sweep"]["n_eff_checksum"] == d2["sweep"]["n_eff_checksum"]     # same table after the gather
    assert d1["sweep"]["solves_rank0"] == 64 and d2["sweep"]["solves_rank0"] == 32   # the work IS sharded (no wall-clock claim)


def test_world_size_mismatch_and_failing_rank_are_errors():
    env = dict(ENV, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    res = _run("--gpus", "2", "--steps", "1", "--warmup", "0", env=env)
    assert res.returncode != 0 and "WORLD_SIZE" in res.stderr
    # no GPU in the build container: without the FAKE switch every rank fails loudly and the launcher reports it
    env = {k: v for k, v in ENV.items() if k != "PLFEM_BENCH_FAKE"}
    import torch
    if not torch.cuda.is_available():
        res = _run("--gpus", "2", "--steps", "1", "--warmup", "0", env=env)
        assert res.returncode != 0 and "exited with code" in res.stderr
