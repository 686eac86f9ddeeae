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
print("all equal:", checked, "frames")
