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


def test_eight_ranks_rehearsal_of_the_sweep():
    """the 8-rank layout of BASELINE configs[3] without hardware -- one line, 8 solves per rank, the
    same table as one rank, every rank analysing with its share of the host's cores.  (A rehearsal of the launcher, the
    partition and the gather under gloo: it says nothing about scaling, and no scaling curve is derived from it.)"""
    one = _run("--gpus", "1", "--steps", "1", "--warmup", "0", "--sweep")
    eight = _run("--gpus", "8", "--steps", "1", "--warmup", "0", "--sweep", timeout=600)
    assert one.returncode == 0 and eight.returncode == 0, one.stderr[-2000:] + eight.stderr[-3000:]
    lines = [l for l in eight.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    d8 = json.loads(lines[0])
    assert d8["n_gpus"] == 8 and d8["scaling"] == "strong" and d8["sweep"]["solves"] == 64
    assert d8["sweep"]["solves_rank0"] == 8
    assert d8["sweep"]["n_eff_checksum"] == d1["sweep"]["n_eff_checksum"]
    ncpu = os.cpu_count() or 64
    assert d8["config"]["host_threads_per_rank"] == max(4, min(16, ncpu // 8))     # Dist: the ranks share the host's cores
    assert d1["config"]["host_threads_per_rank"] is None                            # one rank: the library's default


def test_only_a_rendezvous_failure_is_retried():
    """an import error, an out-of-memory kill or a HIP fault during start-up must not be launched a second
    time; an address-in-use / c10d store error before rank 0 printed anything may."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.is_rendezvous_failure(1, "", "RuntimeError: The server socket has failed to listen on any local network address. "
                                              "port: 29500, useIpv6: false, code: -98, name: EADDRINUSE, message: address already in use")
    assert bench.is_rendezvous_failure(1, "", "torch.distributed.DistNetworkError: The client socket has failed to connect")
    assert not bench.is_rendezvous_failure(1, "", "ModuleNotFoundError: No module named 'scipy'")
    assert not bench.is_rendezvous_failure(-9, "", "EADDRINUSE")                   # killed by a signal: never retried
    assert not bench.is_rendezvous_failure(1, '{"metric": "x"}', "EADDRINUSE")      # rank 0 had already printed its line
    assert not bench.is_rendezvous_failure(1, "", "HIP error: an illegal memory access was encountered")
    # and the failing-rank path of the launcher reports the rank's own stderr, without a second launch
    env = {k: v for k, v in ENV.items() if k != "PLFEM_BENCH_FAKE"}
    import torch
    if not torch.cuda.is_available():
        res = _run("--gpus", "2", "--steps", "1", "--warmup", "0", env=env)
        assert res.returncode != 0 and "retrying" not in res.stderr and "bench.py needs a GPU" in res.stderr


def test_pmc_summaries_are_keyed_by_rung_and_checked_against_the_library(tmp_path, monkeypatch):
    """bench.py read C1's PMC file for every ladder rung (traffic ratios of 5.9 and 0.23), and
    nothing noticed when the committed counters no longer belonged to the code.  pmc_for_levels takes the file of ITS rung,
    latest round first, and refuses one whose kernel regexes / counted kernels are not in the built library."""
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    fam = lambda regex, names: {"families": {"k_spmv_b_block": {"regex": regex, "hbm_bytes": 5.0e7, "kernel_names": names}}, "mfma": {}}
    (prof / "r09_pmc_families_L2.json").write_text(json.dumps(fam("k_spmv_b_block", ["k_spmv_b_block_il<4, 2>"])))
    (prof / "r10_pmc_families_L2.json").write_text(json.dumps(fam("k_kernel_that_is_gone", [])))
    (prof / "r11_pmc_families_L2.json").write_text(json.dumps(fam("k_spmv_b_block", ["k_spmv_b_block_old<4>"])))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "library_kernel_names", lambda: ["k_spmv_b_block_il<4, 2>(int, long, int const*)", "k_fwd<4, 4>(SweepArgs)"])
    fams, _mfma, path, check = bench.pmc_for_levels(2)
    assert path.endswith("r09_pmc_families_L2.json") and check == "ok" and fams["k_spmv_b_block"]["hbm_bytes"] == 5.0e7
    fams, _mfma, path, check = bench.pmc_for_levels(0)          # no pass for this rung: null, never another rung's bytes
    assert fams == {} and path is None and "no PMC pass" in check
    (prof / "r09_pmc_families_L2.json").unlink()
    fams, _mfma, path, check = bench.pmc_for_levels(2)          # only stale files left: refused, with the reason
    assert fams == {} and path is None and ("is not in the built library" in check or "matches no kernel" in check)
    # the committed files of this repository against the library that is built here: accepted with their families, or
    # refused with a reason and NO traffic (a kernel was renamed since the passes were collected: scripts/gpu_pmc_round.sh)
    monkeypatch.undo()
    for levels in (0, 1, 2):
        fams, _mfma, path, check = bench.pmc_for_levels(levels)
        if path is not None:
            assert check.startswith("ok") and len(fams) >= 8, (levels, path, check)
        else:
            assert fams == {} and isinstance(check, str) and check, (levels, check)
            print(f"rung L = {levels}: committed PMC summary refused ({check}): bench.py reports traffic: null until it is regenerated")
