#!/usr/bin/env python3
"""Config-3 / config-4 shaped single-cloud runs on one MI355X:
   python tools/bench_cloud.py --kind multilevel --points 100000000
   python tools/bench_cloud.py --kind loft --points 5000000
   python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/bench_cloud.py --sharded ...
Prints one JSON line with the per-stage HIP-event times and algorithmic GB/s.  --sharded: one
process per GPU, each generates and owns a contiguous index range of the cloud (counter-based RNG:
any sharding yields the same points) and hope_amd.sharding.segment_cloud_sharded runs the path with
its RCCL exchanges; time = max over ranks."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="multilevel", choices=["multilevel", "loft"])
    ap.add_argument("--points", type=int, default=100_000_000)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sharded", action="store_true")
    args = ap.parse_args()
    import numpy as np
    import torch
    from hope_amd import synth
    from hope_amd.capi import HopeContext, HopeLib
    if args.sharded:
        return sharded(args)
    dev = torch.device("cuda", 0)
    if args.kind == "multilevel":
        parts = []
        chunk = 10_000_000
        for i0 in range(0, args.points, chunk):
            parts.append(synth.make_multilevel_cloud(min(chunk, args.points - i0), first_index=i0, device=dev))
        xyz = torch.cat(parts)
        del parts
    else:
        xyz = torch.from_numpy(synth.make_loft_cloud(args.points)).to(dev)
    torch.cuda.synchronize()
    n = xyz.shape[0]
    ctx = HopeContext(HopeLib())
    for _ in range(args.warmup):
        st = ctx.segment_cloud_device(xyz.data_ptr(), n, 12)
    ctx.profile_enable(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        st = ctx.segment_cloud_device(xyz.data_ptr(), n, 12)
    dt = (time.perf_counter() - t0) / args.steps
    c = ctx.counts(0)
    stages = [{"stage": name, "ms": round(ms / args.steps, 4), "GBs": round(by / args.steps / (ms / args.steps * 1e-3) / 1e9, 1) if ms > 0 else 0}
              for name, ms, regions, by in ctx.profile() if regions]
    print(json.dumps({"kind": args.kind, "points": n, "status": st, "ms_per_cloud": dt * 1e3, "points_per_s": n / dt,
                      "n_raw": c.n_raw, "n_vox": c.n_vox, "n_fit": c.n_fit, "n_clusters": c.n_clusters,
                      "n_planes": c.n_planes, "grid": list(c.grid), "xs": ctx.xs_stats(), "stages": stages}))


def sharded(args):
    import torch
    import torch.distributed as dist
    from hope_amd import synth
    from hope_amd.capi import HopeContext, HopeLib
    from hope_amd.sharding import frame_partition, segment_cloud_sharded
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    lo, hi = frame_partition(args.points, world, rank)
    parts, chunk = [], 10_000_000
    for i0 in range(lo, hi, chunk):
        parts.append(synth.make_multilevel_cloud(min(chunk, hi - i0), first_index=i0, device=dev))
    xyz = torch.cat(parts)
    del parts
    ctx = HopeContext(HopeLib(), device=local)
    for _ in range(args.warmup):
        info = segment_cloud_sharded(ctx, xyz, dist)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        info = segment_cloud_sharded(ctx, xyz, dist)
    torch.cuda.synchronize()
    el = torch.tensor([(time.perf_counter() - t0) / args.steps], dtype=torch.float64, device=dev)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    if rank == 0:
        c = ctx.counts(0)
        print(json.dumps({"kind": "multilevel-sharded", "points": args.points, "n_gpus": world,
                          "ms_per_cloud": float(el.item()) * 1e3, "points_per_s": args.points / float(el.item()),
                          "n_raw": c.n_raw, "n_vox": c.n_vox, "n_fit": c.n_fit, "path": info["path"],
                          "n_clusters": info["result"]["n_clusters"], "n_planes": info["result"]["n_planes"],
                          "sent_by_rank0": info["sent"], "received_by_rank0": info["received"]}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
