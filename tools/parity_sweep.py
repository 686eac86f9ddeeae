"""Wider bit-exact sweep than the test-suite (diagnostic; GPU box): many synthetic frames from several
seeds, odd image shapes and thresholds, oracle vs HIP on every stage.
   python tools/parity_sweep.py [frames_per_batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from hope_amd import synth
from hope_amd.capi import HopeConfig, HopeContext, HopeLib
from parity import assert_frame_equal

n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
hip = HopeLib()
orc = HopeLib(os.path.join(ROOT, "oracle", "_build", "libhope_oracle.so"), "hope_oracle_")
t0 = time.time()
checked = 0
for first in (0, 300, 777, 1500, 4000):
    depth, t, q = synth.make_depth_frames(n, first_frame=first, device="cpu")
    depth = depth.numpy().view(np.uint16)
    g, o = HopeContext(hip), HopeContext(orc)
    for c in (g, o):
        c.set_frame_poses(t.numpy(), q.numpy())
        c.segment_depth(depth)
    for f in range(n):
        assert_frame_equal(o, g, f)
        checked += 1
    print("frames", first, "..", first + n, "ok", round(time.time() - t0, 1), "s", flush=True)
# odd shapes: crops of a frame (widths / heights that are no multiples of the 4x16 chunk)
depth, t, q = synth.make_depth_frames(2, first_frame=9, device="cpu")
depth = depth.numpy().view(np.uint16)
for (h, w) in ((480, 640), (479, 637), (121, 333), (5, 17), (3, 640), (480, 15), (64, 64), (33, 47)):
    crop = np.ascontiguousarray(depth[:, :h, :w])
    g, o = HopeContext(hip), HopeContext(orc)
    for c in (g, o):
        c.set_frame_poses(t.numpy(), q.numpy())
        c.segment_depth(crop)
    for f in range(2):
        assert_frame_equal(o, g, f)
        checked += 1
    print("shape", (h, w), "ok", flush=True)
# other thresholds / neighbour counts than the reference's defaults
depth, t, q = synth.make_depth_frames(6, first_frame=2024, device="cpu")
depth = depth.numpy().view(np.uint16)
for kw in (dict(th_xy=0.03, th_z=0.01), dict(th_xy=0.08, th_z=0.02), dict(th_xy=0.05, th_z=0.05),
           dict(th_xy=0.02, th_z=0.02), dict(th_xy=0.1, th_z=0.01), dict(knn=4), dict(knn=12), dict(knn=16),
           dict(th_xy=0.04, th_z=0.015, knn=6), dict(min_depth=0.8, max_depth=2.5), dict(th_xy=0.01, th_z=0.004)):
    g, o = HopeContext(hip, **kw), HopeContext(orc, **kw)
    for c in (g, o):
        c.set_frame_poses(t.numpy(), q.numpy())
        c.segment_depth(depth)
    for f in range(6):
        assert_frame_equal(o, g, f)
        checked += 1
    print("config", kw, "ok", [g.counts(f).n_planes for f in range(6)], round(time.time() - t0, 1), "s", flush=True)
# point clouds: structured scenes at several sizes and seeds, clutter, a lone wall, raw points with
# a coarse leaf (many points per voxel) and a fine one (few), with and without filter + pose
rng = np.random.default_rng(3)
clouds = {
    "loft 150k": synth.make_loft_cloud(150_000, seed=5),
    "loft 400k": synth.make_loft_cloud(400_000, seed=9),
    "multilevel 300k": synth.make_multilevel_cloud(300_000, seed=17, extent=8.0, height=3.0).numpy(),
    "multilevel 500k": synth.make_multilevel_cloud(500_000, first_index=12345, seed=42, extent=10.0, height=4.0).numpy(),
    "clutter 200k": rng.uniform([-2, -2, 0], [2, 2, 2], (200_000, 3)).astype(np.float32),
    "wall 100k": np.stack([rng.uniform(0, 4, 100_000), rng.normal(0, 0.003, 100_000),
                           rng.uniform(0, 2.5, 100_000)], 1).astype(np.float32),
    "slope 100k": (lambda x, y: np.stack([x, y, 0.1 * x + rng.normal(0, 0.002, x.shape)], 1).astype(np.float32))(
        rng.uniform(0, 3, 100_000), rng.uniform(0, 3, 100_000)),
}
for name, xyz in clouds.items():
    for kw, filt in ((dict(), False), (dict(th_xy=0.03, th_z=0.01), False), (dict(th_xy=0.12, th_z=0.03), False),
                     (dict(min_depth=-50.0, max_depth=50.0), True)):
        g, o = HopeContext(hip, **kw), HopeContext(orc, **kw)
        for c in (g, o):
            if filt:
                c.set_pose([0.3, -0.2, 0.1], [0.01, -0.02, 0.3, 0.95])
            c.segment_cloud(xyz, apply_filter_transform=filt)
        assert_frame_equal(o, g, 0)
        checked += 1
    print("cloud", name, "ok", g.counts(0).n_vox, g.counts(0).n_planes, round(time.time() - t0, 1), "s", flush=True)
print("all equal:", checked, "frames")
