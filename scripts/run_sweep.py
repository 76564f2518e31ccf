#!/usr/bin/env python3
"""Multi-band parametric sweep (BASELINE.json config 4): 16 cross-sections x 4 wavelengths = 64
independent vectorial solves sharded over the GPUs of one node, results gathered with one RCCL
all_gather of padded records.

    python scripts/run_sweep.py [--items 64] [--levels 1] [--modes 10]                       # 1 GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29511 scripts/run_sweep.py                                             # 8 GPUs
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=64)
    ap.add_argument("--levels", type=int, default=1)
    ap.add_argument("--refinement", type=float, default=1.0)
    ap.add_argument("--modes", type=int, default=10)
    ap.add_argument("--lanes", type=int, default=1, help="solves in flight per GPU")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from pl_fem_vectoriel_amd.sweep import multiband_sweep_items, run_sweep

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    items = multiband_sweep_items(n_modes=args.modes, mesh_levels=args.levels, mesh_refinement=args.refinement)[:args.items]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    table, n_local = run_sweep(items, rank, world, device=local_rank, lanes=args.lanes)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"solves": len(items), "n_gpus": world, "seconds": dt, "solves_per_s": len(items) / dt,
                          "modes_per_s": len(items) * args.modes / dt, "local_solves_rank0": n_local,
                          "first": {str(i): [round(float(x), 8) for x in table[i][:3]] for i in sorted(table)[:4]}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
