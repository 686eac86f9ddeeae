#!/usr/bin/env python3
"""bench.py — throughput of the HoPE PlaneSegment hot path on MI355X.

A "step" is one pass of the whole path (A1..A12: back-projection, range filter, pose
transform, voxel grid, normals, |nz| gate, 8-NN, z-growing, per-cluster verification) over
one batch of synthetic depth frames that is already resident in HBM when the timed region
starts; the small result set (per-frame counts + cluster records) is copied back inside the
step.  Workload at N=1: BASELINE.json configs[1] — 1024 synthetic 640x480 depth frames
(tabletop + floor) on one MI355X.  With N GPUs every rank processes its own 1024 frames
(frames are independent: no data-path collective, weak scaling).

Usage: python bench.py --gpus N --steps K --warmup W
       (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 640, 480
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md

# The ALGORITHMIC bytes of every timed stage are accounted by the library
# itself from the sizes of each call (hope_profile_stage_bytes), next to the HIP-event time of
# the same region on the library's own stream.


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1024, help="depth frames per GPU per step")
    ap.add_argument("--cpu-frames", type=int, default=512, help="frames timed through the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(depth_np, t_np, q_np):
    """The oracle (CPU restatement of the reference path, single thread like the reference)
    timed on this host's cores over a bounded sample of the same frames."""
    from hope_amd.capi import HopeContext, HopeLib
    lib_path = os.path.join(ROOT, "oracle", "_build", "libhope_oracle.so")
    if not os.path.exists(lib_path):
        return None
    orc = HopeContext(HopeLib(lib_path, "hope_oracle_"))
    orc.set_frame_poses(t_np, q_np)
    t0 = time.perf_counter()
    orc.segment_depth(depth_np)
    dt = time.perf_counter() - t0
    n_raw = sum(orc.counts(f).n_raw for f in range(depth_np.shape[0]))
    whole_ms, ref_ms = orc.timing()
    return {
        "value": n_raw / dt, "unit": "points/s", "cores": 1, "kind": "port",
        "sample": f"first {depth_np.shape[0]} frames of the same synthetic batch, oracle/hope_oracle.cpp "
                  f"-O2 single thread, {dt:.1f} s wall",
        "ms_per_frame": 1e3 * dt / depth_np.shape[0],
        "ref_window_ms_per_frame": ref_ms / depth_np.shape[0],
        "host_cpus_visible": os.cpu_count(),
    }


def main():
    args = parse_args()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from hope_amd import synth
    from hope_amd.capi import HopeContext, HopeLib

    # ---- synthetic batch, generated on the device (config 2; rank r gets frames r*F .. r*F+F-1)
    F = args.frames
    depth = torch.empty((F, H, W), dtype=torch.uint16, device=dev)
    ts, qs = [], []
    chunk = 32
    for f0 in range(0, F, chunk):
        n = min(chunk, F - f0)
        d, t, q = synth.make_depth_frames(n, first_frame=rank * F + f0, device=dev)
        depth[f0:f0 + n] = d
        ts.append(t)
        qs.append(q)
    t_np = torch.cat(ts).numpy()
    q_np = torch.cat(qs).numpy()
    torch.cuda.synchronize()

    ctx = HopeContext(HopeLib(), device=local_rank)
    ctx.set_frame_poses(t_np, q_np)
    ptr = depth.data_ptr()

    def step():
        st = ctx.segment_depth_device(ptr, W, H, F)
        assert st >= 0

    for _ in range(args.warmup):
        step()
    ctx.profile_enable(True)  # HIP events on the library's own stream, recorded inside the timed steps
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # each call ends with a stream synchronise (results are on the host)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    n_raw = sum(ctx.counts(f).n_raw for f in range(F))
    n_vox = sum(ctx.counts(f).n_vox for f in range(F))
    n_fit = sum(ctx.counts(f).n_fit for f in range(F))
    n_planes = sum(ctx.counts(f).n_planes for f in range(F))
    tot = torch.tensor([n_raw, n_vox, n_fit, n_planes], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    tot_raw = float(tot[0].item())

    if rank == 0:
        prof = ctx.profile()
        stages = []
        for name, ms, regions, alg_bytes in prof:
            if regions == 0:
                continue
            per_step_ms = ms / args.steps
            gbs = alg_bytes / args.steps / (per_step_ms * 1e-3) / 1e9 if per_step_ms > 0 else 0.0
            stages.append({"stage": name, "ms_per_step": round(per_step_ms, 4),
                           "timed_regions_per_step": regions / args.steps,
                           "alg_bytes_per_step": alg_bytes / args.steps, "achieved_GBs": round(gbs, 1)})
        dom = max(stages, key=lambda s: s["ms_per_step"])
        dom_launch_ms = dom["ms_per_step"] / dom["timed_regions_per_step"]
        dom_bytes = dom["alg_bytes_per_step"] / dom["timed_regions_per_step"]
        achieved = dom_bytes / (dom_launch_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                # recorded per bench step (tools/pmc_traffic.py); reported per launch like `achieved`
                traffic = json.load(open(tpath)).get(dom["stage"])
                if traffic is not None:
                    traffic = traffic / dom["timed_regions_per_step"]
            except Exception:
                traffic = None
        out = {
            "metric": "points/sec through PlaneSegment",
            "value": tot_raw * args.steps / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{F} synthetic 640x480 uint16 depth frames per GPU (tabletop+floor, "
                                   "BASELINE.json configs[1]), th_xy=0.05 th_z=0.02, frames sharded across GPUs",
                       "frames_per_gpu": F, "parallelism": f"frames x{world}"},
            "ms_per_frame": 1e3 * elapsed / args.steps / F,
            "pixels_per_s": world * F * W * H * args.steps / elapsed,
            "points_raw_per_step": tot_raw, "points_downsampled": float(tot[1].item()),
            "points_normal_fit": float(tot[2].item()), "planes": float(tot[3].item()),
            "roofline": {"bound": "hbm", "kernel": dom["stage"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "avg_launch_ms": dom_launch_ms},
            "stages": stages,
        }
        if not args.no_cpu_baseline and world == 1:
            n = min(args.cpu_frames, F)
            out["cpu_baseline"] = cpu_baseline(depth[:n].cpu().numpy().view(np.uint16), t_np[:n], q_np[:n])
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
