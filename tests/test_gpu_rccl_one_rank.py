"""The RCCL calls of the multi-GPU path on a ONE-GPU box: a process group of one rank (backend nccl = RCCL), so that
``init_process_group(device_id=...)``, the barrier, the float64 MAX all-reduce of ``bench.py`` and the all-gather of the
sweep's records on device tensors have run through RCCL at least once under the driver (the nccl branch had
only ever executed under gloo).  One rank says nothing about scaling; it catches a wrong dtype / device / API use."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               PLFEM_BENCH_FORCE_DIST="1")
    for k in ("PLFEM_BENCH_FAKE", "PLFEM_BENCH_BACKEND", "PLFEM_BENCH_SAME_DEVICE"):
        env.pop(k, None)
    return env


@pytest.mark.gpu
def test_one_rank_process_group_runs_the_collectives_through_rccl(gpu_device, built_library):
    # the headline configuration: barrier + all_reduce(MAX) of the elapsed time on a device tensor
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline"], env=_env(), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and 5.0 < d["ms_per_step"] < 200.0
    # the sweep: the all-gather of the 64 records on device tensors, same table as without a process group
    plain_env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "PLFEM_BENCH_FORCE_DIST")}
    flags = ["--gpus", "1", "--sweep", "--steps", "1", "--warmup", "0", "--lanes", "4"]
    plain = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=plain_env, capture_output=True, text=True,
                           timeout=900)
    grouped = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=_env(), capture_output=True, text=True,
                             timeout=900)
    assert plain.returncode == 0 and grouped.returncode == 0, plain.stderr[-2000:] + grouped.stderr[-3000:]
    d0 = json.loads([l for l in plain.stdout.splitlines() if l.startswith("{")][-1])
    d1 = json.loads([l for l in grouped.stdout.splitlines() if l.startswith("{")][-1])
    assert d1["sweep"]["solves"] == 64 and d1["sweep"]["solves_rank0"] == 64
    assert d1["sweep"]["n_eff_checksum"] == d0["sweep"]["n_eff_checksum"]
    assert "backend nccl" in d1["config"]["parallelism"]
