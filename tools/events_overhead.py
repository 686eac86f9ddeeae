#!/usr/bin/env python3
"""What the per-stage HIP events of the bench cost: the 1024-frame step with and without hope_profile_enable."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hope_amd import synth  # noqa: E402
from hope_amd.capi import HopeContext, HopeLib  # noqa: E402

F, W, H = 1024, 640, 480
dev = torch.device("cuda:0")
ds, ts, qs = [], [], []
for f0 in range(0, F, 128):
    d, t, q = synth.make_depth_frames(min(128, F - f0), first_frame=f0, device=dev)
    ds.append(d); ts.append(t.cpu()); qs.append(q.cpu())
depth = torch.cat(ds)
ctx = HopeContext(HopeLib())
ctx.set_frame_poses(torch.cat(ts).numpy(), torch.cat(qs).numpy())
torch.cuda.synchronize()
for on in (False, True, False, True):
    ctx.profile_enable(on)
    for _ in range(2):
        ctx.segment_depth_device(depth.data_ptr(), W, H, F)
    t0 = time.perf_counter()
    for _ in range(5):
        ctx.segment_depth_device(depth.data_ptr(), W, H, F)
    print("stage events %s: %.3f ms per step" % ("on " if on else "off", (time.perf_counter() - t0) / 5 * 1e3))
