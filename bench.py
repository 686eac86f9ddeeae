#!/usr/bin/env python3
"""bench.py — throughput of the HoPE PlaneSegment hot path on MI355X.

A "step" is one pass of the whole path (A1..A12: back-projection, range filter, pose
transform, voxel grid, normals, |nz| gate, 8-NN, z-growing, per-cluster verification) over
one batch of synthetic input that is already resident in HBM when the timed region starts;
the small result set (per-frame counts + cluster records) is copied back inside the step.

--workload frames (default; BASELINE.json configs[1], the config the metric is quoted on):
    1024 synthetic 640x480 depth frames (tabletop + floor) per GPU.  With N GPUs every rank
    processes its own 1024 frames (frames are independent: no data-path collective, weak scaling).
--workload cloud (BASELINE.json configs[3], the north_star's 10^8-point multi-level scene):
    one cloud of --points points, generated on the device.  With N GPUs the cloud is sharded
    by point range and hope_amd.sharding.segment_cloud_sharded runs the path with its RCCL
    exchanges (bounding box + int64 z-histogram all-reduce, one all-to-all of routed points,
    all-gathers of the down-sampled cloud, the normals and the k-NN rows): strong scaling.

Usage: python bench.py --gpus N --steps K --warmup W [--workload cloud]
       (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

W, H = 640, 480
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libhope_oracle.so")

# The ALGORITHMIC bytes of every timed stage are accounted by the library
# itself from the sizes of each call (hope_profile_stage_bytes), next to the HIP-event time of
# the same region on the library's own stream.


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="frames", choices=["frames", "cloud"])
    ap.add_argument("--frames", type=int, default=1024, help="frames workload: depth frames per GPU per step")
    ap.add_argument("--points", type=int, default=100_000_000, help="cloud workload: points of the scene (all GPUs)")
    ap.add_argument("--cpu-frames", type=int, default=256, help="frames timed through the CPU oracle on one core")
    ap.add_argument("--cpu-points", type=int, default=2_000_000, help="cloud workload: points of the oracle's sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hip-lib", default=None, help="diagnostics only: another build of libhope_hip.so (tools/ablate.sh)")
    return ap.parse_args()


def host_threads():
    """Worker threads for the all-cores leg: the CPUs this process may run on, at most 16 (a one-GPU
    box's share; os.cpu_count() shows the whole machine there)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(16, n))


def cpu_baseline_frames(depth_np, t_np, q_np, n_single):
    """The oracle (CPU restatement of the reference path) timed on this host's cores over a bounded
    sample of the same frames: (a) one core, like the single-threaded reference; (b) all cores, one
    oracle context per thread over a slice of the frames (frames are independent; the oracle's C
    calls release the GIL)."""
    from hope_amd.capi import HopeContext, HopeLib
    if not os.path.exists(ORACLE_LIB):
        return None
    lib = HopeLib(ORACLE_LIB, "hope_oracle_")
    n_single = min(n_single, depth_np.shape[0])
    orc = HopeContext(lib)
    orc.set_frame_poses(t_np[:n_single], q_np[:n_single])
    t0 = time.perf_counter()
    orc.segment_depth(depth_np[:n_single])
    dt = time.perf_counter() - t0
    n_raw = sum(orc.counts(f).n_raw for f in range(n_single))
    whole_ms, ref_ms = orc.timing()
    # (b) all cores
    nt = host_threads()
    F = depth_np.shape[0]
    bounds = [F * i // nt for i in range(nt + 1)]
    ctxs = [HopeContext(lib) for _ in range(nt)]
    raw = [0] * nt

    def work(i):
        lo, hi = bounds[i], bounds[i + 1]
        if hi > lo:
            ctxs[i].set_frame_poses(t_np[lo:hi], q_np[lo:hi])
            ctxs[i].segment_depth(depth_np[lo:hi])
            raw[i] = sum(ctxs[i].counts(f).n_raw for f in range(hi - lo))

    th = [threading.Thread(target=work, args=(i,)) for i in range(nt)]
    t1 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt_all = time.perf_counter() - t1
    return {
        "value": n_raw / dt, "unit": "points/s", "cores": 1, "kind": "port",
        "sample": f"first {n_single} frames of the same synthetic batch, oracle/hope_oracle.cpp "
                  f"-O2 single thread, {dt:.1f} s wall",
        "ms_per_frame": 1e3 * dt / n_single,
        "ref_window_ms_per_frame": ref_ms / n_single,
        "all_cores": {"value": sum(raw) / dt_all, "unit": "points/s", "cores": nt,
                      "sample": f"{F} frames of the same batch, one oracle context per thread over a slice of "
                                f"the frames, {dt_all:.1f} s wall",
                      "ms_per_frame": 1e3 * dt_all / F},
        "host_cpus_visible": os.cpu_count(),
    }


def cpu_baseline_cloud(xyz_np):
    from hope_amd.capi import HopeContext, HopeLib
    if not os.path.exists(ORACLE_LIB):
        return None
    orc = HopeContext(HopeLib(ORACLE_LIB, "hope_oracle_"))
    t0 = time.perf_counter()
    orc.segment_cloud(xyz_np)
    dt = time.perf_counter() - t0
    whole_ms, ref_ms = orc.timing()
    return {
        "value": xyz_np.shape[0] / dt, "unit": "points/s", "cores": 1, "kind": "port",
        "sample": f"the first {xyz_np.shape[0]} points of the same scene (same density per surface: a "
                  f"{xyz_np.shape[0]}-point scene of the same layout), oracle/hope_oracle.cpp -O2 single thread, "
                  f"{dt:.1f} s wall; one cloud is one extraction problem, so there is no all-cores leg",
        "ref_window_ms": ref_ms, "host_cpus_visible": os.cpu_count(),
    }


def stage_report(prof, steps, traffic_file):
    stages = []
    for name, ms, regions, alg_bytes in prof:
        if regions == 0:
            continue
        per_step_ms = ms / steps
        gbs = alg_bytes / steps / (per_step_ms * 1e-3) / 1e9 if per_step_ms > 0 else 0.0
        stages.append({"stage": name, "ms_per_step": round(per_step_ms, 4),
                       "timed_regions_per_step": regions / steps,
                       "alg_bytes_per_step": alg_bytes / steps, "achieved_GBs": round(gbs, 1)})
    dom = max(stages, key=lambda s: s["ms_per_step"])
    dom_launch_ms = dom["ms_per_step"] / dom["timed_regions_per_step"]
    dom_bytes = dom["alg_bytes_per_step"] / dom["timed_regions_per_step"]
    achieved = dom_bytes / (dom_launch_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", traffic_file)
    if os.path.exists(tpath):
        try:
            # recorded per bench step (tools/pmc_traffic.py); reported per launch like `achieved`
            traffic = json.load(open(tpath)).get(dom["stage"])
            if traffic is not None:
                traffic = traffic / dom["timed_regions_per_step"]
        except Exception:
            traffic = None
    whole_bytes = sum(s["alg_bytes_per_step"] for s in stages)
    roofline = {"bound": "hbm", "kernel": dom["stage"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_ms": dom_launch_ms, "alg_bytes_per_launch": dom_bytes,
                "whole_step_alg_bytes": whole_bytes}
    return stages, roofline


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

    ctx = HopeContext(HopeLib(args.hip_lib) if args.hip_lib else HopeLib(), device=local_rank)

    if args.workload == "frames":
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
        ctx.set_frame_poses(t_np, q_np)
        ptr = depth.data_ptr()

        def step():
            st = ctx.segment_depth_device(ptr, W, H, F)
            assert st >= 0
        n_units = F
    else:
        # ---- one multi-level scene (config 4), generated on the device; rank r owns a contiguous index range
        from hope_amd.sharding import frame_partition, segment_cloud_sharded
        lo, hi = frame_partition(args.points, world, rank)
        parts, chunk = [], 10_000_000
        for i0 in range(lo, hi, chunk):
            parts.append(synth.make_multilevel_cloud(min(chunk, hi - i0), first_index=i0, device=dev))
        xyz = torch.cat(parts) if parts else torch.zeros((0, 3), dtype=torch.float32, device=dev)
        del parts
        torch.cuda.synchronize()
        info = {}

        def step():
            if world == 1:
                st = ctx.segment_cloud_device(xyz.data_ptr(), xyz.shape[0], 12)
                assert st >= 0
            else:
                info.update(segment_cloud_sharded(ctx, xyz, dist))
        n_units = 1

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

    nf = ctx.num_frames()
    n_raw = sum(ctx.counts(f).n_raw for f in range(nf))
    n_vox = sum(ctx.counts(f).n_vox for f in range(nf))
    n_fit = sum(ctx.counts(f).n_fit for f in range(nf))
    n_planes = sum(ctx.counts(f).n_planes for f in range(nf))
    tot = torch.tensor([n_raw, n_vox, n_fit, n_planes], dtype=torch.float64, device=dev)
    if world > 1 and args.workload == "frames":
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)   # (sharded cloud: every rank already holds the whole result)
    tot_raw = float(tot[0].item())
    if args.workload == "cloud" and world > 1 and rank == 0 and info.get("result") is not None:
        tot[3] = float(info["result"]["n_planes"])   # (A9..A12 per z band: rank 0 holds the merged records)

    if rank == 0:
        stages, roofline = stage_report(ctx.profile(), args.steps,
                                        "hbm_traffic.json" if args.workload == "frames" else "hbm_traffic_cloud.json")
        out = {
            "metric": "points/sec through PlaneSegment",
            "value": tot_raw * args.steps / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak" if args.workload == "frames" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
        }
        if args.workload == "frames":
            F = args.frames
            out["config"] = {"workload": f"{F} synthetic 640x480 uint16 depth frames per GPU (tabletop+floor, "
                                         "BASELINE.json configs[1]), th_xy=0.05 th_z=0.02, frames sharded across GPUs",
                             "frames_per_gpu": F, "parallelism": f"frames x{world}"}
            out["ms_per_frame"] = 1e3 * elapsed / args.steps / F
            out["pixels_per_s"] = world * F * W * H * args.steps / elapsed
        else:
            out["config"] = {"workload": f"one synthetic multi-level scene of {args.points} points (30 m x 30 m x 6 m, "
                                         "shelves/tables + clutter, 2 mm noise; BASELINE.json configs[3]), th_xy=0.05 "
                                         "th_z=0.02, POINT_CLOUD mode (no filter, no pose)",
                             "points": args.points,
                             "parallelism": "one GPU" if world == 1 else
                                            f"point ranges x{world}, z-band exchange over RCCL (hope_amd/sharding.py)"}
            if world > 1:
                out["exchange"] = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in info.items()
                                   if k in ("sent", "received", "own_vox", "own_fit", "path")}
        out.update({"points_raw_per_step": tot_raw, "points_downsampled": float(tot[1].item()),
                    "points_normal_fit": float(tot[2].item()), "planes": float(tot[3].item()),
                    "roofline": roofline, "stages": stages})
        if not args.no_cpu_baseline and world == 1:
            if args.workload == "frames":
                out["cpu_baseline"] = cpu_baseline_frames(depth.cpu().numpy().view(np.uint16), t_np, q_np, args.cpu_frames)
            else:
                sample = synth.make_multilevel_cloud(min(args.cpu_points, args.points), first_index=0, device=dev)
                out["cpu_baseline"] = cpu_baseline_cloud(sample.cpu().numpy())
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
