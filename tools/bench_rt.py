#!/usr/bin/env python3
"""Config 5: the hope_ros streaming loop without ROS.  Synthetic config-2
frames are replayed one at a time as host point clouds (the cloud_topic payload) through the RT entry
(hope_segment_cloud_xyz with filter + pose, HOPE_MODE_RT, height window 0.8..1.5 from cfg/hope.cfg,
changed mid-stream to 0.0..0.3 like a dynamic_reconfigure call).  Latency per frame = H2D of the
cloud + kernels + D2H of the result, measured around the C-ABI call; prints p50 / p99 in ms."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    args = ap.parse_args()
    import numpy as np
    from hope_amd import synth
    from hope_amd.capi import HOPE_MODE_RT, HopeContext, HopeLib
    depth, t, q = synth.make_depth_frames(32, first_frame=0, device="cpu")
    depth = depth.numpy().view(np.uint16)
    # sensor-frame clouds exactly as GetCloud would hand them over (x, y, z; invalid pixels NaN)
    cols = (np.arange(640, dtype=np.float32) - np.float32(318.6))[None, :]
    rows = (np.arange(480, dtype=np.float32) - np.float32(255.3))[:, None]
    clouds = []
    for f in range(depth.shape[0]):
        z = depth[f].astype(np.float32) / np.float32(5000.0)
        ok = (z < 8.0) & (z > 0.3)
        x = cols * z / np.float32(517.0)
        y = rows * z / np.float32(517.0)
        c = np.stack([x, y, z], -1).astype(np.float32)
        c[~ok] = np.nan
        clouds.append(np.ascontiguousarray(c.reshape(-1, 3)))
    ctx = HopeContext(HopeLib(), mode=HOPE_MODE_RT, min_height=0.8, max_height=1.5)
    lat, found = [], 0
    for i in range(args.frames + 10):
        f = i % len(clouds)
        if i == (args.frames + 10) // 2:
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
    print(json.dumps({"config": "hope_ros streaming loop stand-in, 640x480 clouds (307200 points, 3.7 MB H2D per frame)",
                      "frames": len(lat), "p50_ms": float(np.percentile(lat, 50)), "p99_ms": float(np.percentile(lat, 99)),
                      "mean_ms": float(lat.mean()), "max_ms": float(lat.max()), "frames_with_plane": found,
                      "reference_rate_hz": 30.0}))


if __name__ == "__main__":
    main()
