#!/usr/bin/env python3
"""Host-side time marks of one 1024-frame step (HOPE_DEBUG=1 HOPE_DEBUG_TIMELINE=1): where the host waits and how long
the GPU is left without work between two calls."""
import os
import sys
import time

os.environ["HOPE_DEBUG"] = "1"
os.environ["HOPE_DEBUG_TIMELINE"] = "1"
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
for i in range(4):
    t0 = time.perf_counter()
    ctx.segment_depth_device(depth.data_ptr(), W, H, F)
    print("call %d: %.3f ms" % (i, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
import numpy as np
cells = np.array([np.prod(np.array(ctx.counts(f).grid, dtype=np.int64)) for f in range(F)])
nv = np.array([ctx.counts(f).n_vox for f in range(F)])
print("cells per frame: min %d median %d max %d; voxels per frame: median %d max %d" %
      (cells.min(), np.median(cells), cells.max(), np.median(nv), nv.max()), file=sys.stderr)
ncl = np.array([ctx.counts(f).n_clusters for f in range(F)])
nseg = np.array([ctx.counts(f).n_segments for f in range(F)])
sizes = np.concatenate([ctx.clusters(f)['size'] for f in range(0, F, 16)])
print("clusters per frame: median %d max %d total %d; segments total %d; cluster sizes (every 16th frame): n %d median %d p90 %d max %d, <=64: %d, <=384: %d" %
      (np.median(ncl), ncl.max(), ncl.sum(), nseg.sum(), len(sizes), np.median(sizes), np.percentile(sizes, 90), sizes.max(), (sizes <= 64).sum(), (sizes <= 384).sum()), file=sys.stderr)
