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

The default line is the frames headline plus two sub-objects measured in the same run, by the same rules
(W warm-up steps, K timed steps between barriers and device synchronisations, MAX over ranks):
  "cloud":  the --workload cloud measurement (ms_per_step, value, roofline of ITS dominant kernel, stages,
            cpu_baseline; with N > 1: scaling "strong", exchange.path, bytes sent per rank, ranks_seen);
  "stream": BASELINE.json configs[4], the hope_ros loop without ROS: 640x480 clouds one at a time through the RT
            entry with a height window that changes mid-stream, latency p50/p99 per frame, H2D included (N = 1).

Usage: python bench.py --gpus N --steps K --warmup W [--workload cloud] [--no-cloud] [--no-stream]
  N > 1 under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment): one rank per GPU.
  N > 1 WITHOUT those variables: bench.py starts the N ranks itself, as child processes, before anything in
  this process touches the GPU, and relays rank 0's line.
  Test knobs (one-GPU box): HOPE_BENCH_RANKS_ON_GPU0=1 maps every rank to device 0 and HOPE_BENCH_BACKEND=gloo
  carries the collectives over gloo with host tensors (RCCL refuses two ranks on one device).
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
    ap.add_argument("--cpu-points", type=int, default=20_000_000, help="cloud workload: points of the oracle's sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cloud", action="store_true", help="frames workload: leave out the cloud sub-object")
    ap.add_argument("--no-two-ranges", action="store_true",
                    help="frames workload: leave out the two_ranges sub-object (the batch again with sub_batches = 2)")
    ap.add_argument("--no-stream", action="store_true", help="frames workload: leave out the stream sub-object")
    ap.add_argument("--stream-frames", type=int, default=300)
    ap.add_argument("--sub-batches", type=int, default=0,
                    help="frames workload: hope_config.sub_batches (frame ranges on streams of their own; the per-stage "
                         "times then overlap and describe no kernel running alone)")
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


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start the N ranks as child processes (fresh interpreters; this process has
    not imported torch or touched HIP), wait for them, relay rank 0's JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(args.gpus), "LOCAL_WORLD_SIZE": str(args.gpus),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


class Job:
    """What every measurement needs: the rank, its device, the process group, the timing rule."""

    def __init__(self, args):
        import torch
        import torch.distributed as dist
        self.args, self.torch, self.dist = args, torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.backend = os.environ.get("HOPE_BENCH_BACKEND", "nccl")
        self.dev_index = 0 if os.environ.get("HOPE_BENCH_RANKS_ON_GPU0") else self.local_rank
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        torch.cuda.set_device(self.dev_index)
        self.dev = torch.device("cuda", self.dev_index)
        # where the tensors of the collectives live: the device with RCCL, the host with the gloo test knob
        self.cdev = self.dev if self.backend == "nccl" else torch.device("cpu")
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)

    def new_ctx(self, **kw):
        from hope_amd.capi import HopeContext, HopeLib
        return HopeContext(HopeLib(self.args.hip_lib) if self.args.hip_lib else HopeLib(), device=self.dev_index, **kw)

    def timed(self, ctx, step):
        """W warm-up steps, then exactly K steps between a barrier + device synchronise on both sides; MAX over ranks."""
        torch, dist = self.torch, self.dist
        for _ in range(self.args.warmup):
            step()
        ctx.profile_enable(True)  # HIP events on the library's own stream, recorded inside the timed steps
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(self.args.steps):
            step()  # each call ends with a stream synchronise (results are on the host)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            dist.barrier()
            el = torch.tensor([elapsed], dtype=torch.float64, device=self.cdev)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            elapsed = float(el.item())
        return elapsed   # (the stage accumulators stay as they are: profile_enable() would clear them)

    def totals(self, ctx, reduce_over_ranks):
        torch, dist = self.torch, self.dist
        nf = ctx.num_frames()
        c = [ctx.counts(f) for f in range(nf)]
        tot = torch.tensor([sum(x.n_raw for x in c), sum(x.n_vox for x in c), sum(x.n_fit for x in c),
                            sum(x.n_planes for x in c)], dtype=torch.float64, device=self.cdev)
        if self.world > 1 and reduce_over_ranks:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        return [float(v) for v in tot.tolist()]


def traffic_source(traffic_file):
    """Where roofline.traffic comes from: the committed PMC summary and the build it was taken on."""
    meta = os.path.join(ROOT, "profiles", traffic_file.replace(".json", ".meta.json"))
    src = {"file": "profiles/" + traffic_file, "measured_in_this_run": False}
    if os.path.exists(meta):
        try:
            src.update(json.load(open(meta)))
        except Exception:
            pass
    return src


def run_frames(job):
    """BASELINE.json configs[1]: F synthetic depth frames per GPU (rank r gets frames r*F .. r*F+F-1), weak scaling,
    no data-path collective."""
    import numpy as np
    from hope_amd import synth
    torch, args = job.torch, job.args
    ctx = job.new_ctx(sub_batches=args.sub_batches)
    F = args.frames
    depth = torch.empty((F, H, W), dtype=torch.uint16, device=job.dev)
    ts, qs = [], []
    chunk = 32
    for f0 in range(0, F, chunk):
        n = min(chunk, F - f0)
        d, t, q = synth.make_depth_frames(n, first_frame=job.rank * F + f0, device=job.dev)
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

    elapsed = job.timed(ctx, step)
    tot = job.totals(ctx, True)
    prof = ctx.profile()
    ctx.close()   # (its workspace goes back to the device before the second context below builds its own)
    el2 = None
    if not args.no_two_ranges and args.sub_batches <= 1 and F >= 128:
        # the same batch as two ranges of frames in flight (hope_config.sub_batches = 2): faster, but a kernel's
        # HIP-event duration then includes the other range's kernels, so the headline and its roofline stay on one
        # range.  (Every rank takes part: timed() holds barriers.)
        ctx2 = job.new_ctx(sub_batches=2)
        ctx2.set_frame_poses(t_np, q_np)

        def step2():
            assert ctx2.segment_depth_device(ptr, W, H, F) >= 0

        el2 = job.timed(ctx2, step2)
        ctx2.close()
    if job.rank != 0:
        return None
    stages, roofline = stage_report(prof, args.steps, "hbm_traffic.json")
    roofline["traffic_source"] = traffic_source("hbm_traffic.json")
    out = {
        "value": tot[0] * args.steps / elapsed, "unit": "points/s", "ms_per_step": 1e3 * elapsed / args.steps,
        "scaling": "weak",
        "config": {"workload": f"{F} synthetic 640x480 uint16 depth frames per GPU (tabletop+floor, "
                               "BASELINE.json configs[1]), th_xy=0.05 th_z=0.02, frames sharded across GPUs",
                   "frames_per_gpu": F, "parallelism": f"frames x{job.world}"},
        "ms_per_frame": 1e3 * elapsed / args.steps / F,
        "pixels_per_s": job.world * F * W * H * args.steps / elapsed,
        "points_raw_per_step": tot[0], "points_downsampled": tot[1], "points_normal_fit": tot[2], "planes": tot[3],
        "roofline": roofline, "stages": stages,
    }
    if el2 is not None:
        out["two_ranges"] = {"sub_batches": 2, "ms_per_step": 1e3 * el2 / args.steps,
                             "value": tot[0] * args.steps / el2, "unit": "points/s",
                             "note": "same steps, warm-up and timing; opt-in (doubles the arena)"}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = (cpu_baseline_frames(depth.cpu().numpy().view(np.uint16), t_np, q_np, args.cpu_frames)
                               if job.world == 1 else None)
    return out


def run_cloud(job):
    """BASELINE.json configs[3]: one multi-level scene generated on the device; rank r owns a contiguous index
    range, the path runs with its exchanges (hope_amd/sharding.py): strong scaling."""
    from hope_amd import synth
    from hope_amd.sharding import frame_partition, segment_cloud_sharded
    torch, args, dist = job.torch, job.args, job.dist
    ctx = job.new_ctx()
    lo, hi = frame_partition(args.points, job.world, job.rank)
    parts, chunk = [], 10_000_000
    for i0 in range(lo, hi, chunk):
        parts.append(synth.make_multilevel_cloud(min(chunk, hi - i0), first_index=i0, device=job.dev))
    xyz = torch.cat(parts) if parts else torch.zeros((0, 3), dtype=torch.float32, device=job.dev)
    del parts
    if job.world > 1 and job.cdev.type == "cpu":
        xyz = xyz.cpu()
    torch.cuda.synchronize()
    info = {}

    def step():
        if job.world == 1:
            st = ctx.segment_cloud_device(xyz.data_ptr(), xyz.shape[0], 12)
            assert st >= 0
        else:
            info.update(segment_cloud_sharded(ctx, xyz, dist))

    elapsed = job.timed(ctx, step)
    tot = job.totals(ctx, False)   # (sharded cloud: every rank already holds the whole cloud's A1..A8 counts)
    if job.world > 1 and job.rank == 0 and info.get("result") is not None:
        tot[3] = float(info["result"]["n_planes"])   # (A9..A12 per z band: rank 0 holds the merged records)
    if job.rank != 0:
        return None
    stages, roofline = stage_report(ctx.profile(), args.steps, "hbm_traffic_cloud.json")
    roofline["traffic_source"] = traffic_source("hbm_traffic_cloud.json")
    n_raw = float(args.points) if job.world > 1 else tot[0]
    out = {
        "value": n_raw * args.steps / elapsed, "unit": "points/s", "ms_per_step": 1e3 * elapsed / args.steps,
        "scaling": "strong",
        "config": {"workload": f"one synthetic multi-level scene of {args.points} points (30 m x 30 m x 6 m, "
                               "shelves/tables + clutter, 2 mm noise; BASELINE.json configs[3]), th_xy=0.05 "
                               "th_z=0.02, POINT_CLOUD mode (no filter, no pose)",
                   "points": args.points,
                   "parallelism": "one GPU" if job.world == 1 else
                                  f"point ranges x{job.world}, z-band exchange over "
                                  f"{'RCCL' if job.backend == 'nccl' else job.backend} (hope_amd/sharding.py)"},
        "points_raw_per_step": n_raw, "points_downsampled": tot[1], "points_normal_fit": tot[2], "planes": tot[3],
        # the whole step against the chip: SURVEY.md section 8d's end-to-end bytes (40 B per raw point + 184 B per
        # down-sampled point) over the step time, as a fraction of the HBM peak
        "whole_step": {"alg_bytes": 40.0 * n_raw + 184.0 * tot[1],
                       "achieved": (40.0 * n_raw + 184.0 * tot[1]) / (elapsed / args.steps) / 1e9, "unit": "GB/s",
                       "frac": (40.0 * n_raw + 184.0 * tot[1]) / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS},
        "roofline": roofline, "stages": stages,
    }
    if job.world > 1:
        sent = info.get("sent", [])
        out["exchange"] = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in info.items()
                           if k in ("sent", "received", "own_vox", "own_fit", "path")}
        out["exchange"]["bytes_sent_by_rank0_in_the_all_to_all"] = 12 * int(sum(sent)) if len(sent) else 0
        out["exchange"]["ranks_seen"] = dist.get_world_size()
        out["exchange"]["backend"] = "rccl" if job.backend == "nccl" else job.backend
    if not args.no_cpu_baseline:
        if job.world == 1:
            sample = synth.make_multilevel_cloud(min(args.cpu_points, args.points), first_index=0, device=job.dev)
            out["cpu_baseline"] = cpu_baseline_cloud(sample.cpu().numpy())
        else:
            out["cpu_baseline"] = None
    return out


def run_stream(job):
    """BASELINE.json configs[4]: synthetic config-2 frames replayed one at a time as
    HOST point clouds (the cloud_topic payload) through the RT entry (filter + pose, HOPE_MODE_RT, height window
    0.8..1.5 from cfg/hope.cfg, changed mid-stream to 0.0..0.3 like a dynamic_reconfigure call).  Latency per frame
    = H2D of the cloud + kernels + D2H of the result, around the C-ABI call."""
    import numpy as np
    from hope_amd import synth
    from hope_amd.capi import HOPE_MODE_RT
    n_frames = job.args.stream_frames
    depth, t, q = synth.make_depth_frames(32, first_frame=0, device="cpu")
    depth = depth.numpy().view(np.uint16)
    cols = (np.arange(W, dtype=np.float32) - np.float32(318.6))[None, :]
    rows = (np.arange(H, dtype=np.float32) - np.float32(255.3))[:, None]
    clouds = []
    for f in range(depth.shape[0]):
        z = depth[f].astype(np.float32) / np.float32(5000.0)
        ok = (z < 8.0) & (z > 0.3)
        c = np.stack([cols * z / np.float32(517.0), rows * z / np.float32(517.0), z], -1).astype(np.float32)
        c[~ok] = np.nan
        clouds.append(np.ascontiguousarray(c.reshape(-1, 3)))
    ctx = job.new_ctx(mode=HOPE_MODE_RT, min_height=0.8, max_height=1.5)
    lat, found = [], 0
    for i in range(n_frames + 10):
        f = i % len(clouds)
        if i == (n_frames + 10) // 2:
            ctx.set_height_window(0.0, 0.3)      # configCallback mid-stream
        ctx.set_pose(t[f].numpy(), q[f].numpy())
        t0 = time.perf_counter()
        ctx.segment_cloud(clouds[f], apply_filter_transform=True)
        m = ctx.max_plane(0)
        dt = time.perf_counter() - t0
        if i >= 10:
            lat.append(dt * 1e3)
            found += int(m.valid)
    lat = np.array(lat)
    return {"config": {"workload": "hope_ros streaming loop stand-in (BASELINE.json configs[4]): 640x480 host clouds "
                                   "(307200 points, 3.7 MB H2D per frame) one at a time through the RT entry, height "
                                   "window 0.8..1.5 changed to 0.0..0.3 mid-stream"},
            "frames": len(lat), "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
            "mean_ms": float(lat.mean()), "max_ms": float(lat.max()), "frames_with_plane": found,
            "higher_is_better": False, "unit": "ms per frame (H2D + kernels + D2H)", "reference_rate_hz": 30.0}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    job = Job(args)
    head = run_frames(job) if args.workload == "frames" else run_cloud(job)
    cloud = stream = None
    if args.workload == "frames":
        if not args.no_cloud:
            cloud = run_cloud(job)
        if not args.no_stream and job.world == 1:
            stream = run_stream(job)
    if job.rank == 0:
        out = {"metric": "points/sec through PlaneSegment", "value": head["value"], "unit": "points/s",
               "n_gpus": job.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": head["ms_per_step"],
               "higher_is_better": True, "scaling": head["scaling"], "vs_baseline": None, "dtype": "f32",
               "data": "synthetic"}
        out.update({k: v for k, v in head.items() if k not in out})
        if cloud is not None:
            cloud.update({"n_gpus": job.world, "steps": args.steps, "warmup": args.warmup, "dtype": "f32",
                          "data": "synthetic", "higher_is_better": True})
            out["cloud"] = cloud
        if stream is not None:
            out["stream"] = stream
        print(json.dumps(out), flush=True)
    if job.world > 1:
        job.dist.barrier()
        job.dist.destroy_process_group()


if __name__ == "__main__":
    main()
