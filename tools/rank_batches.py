"""The batches that ranks 1..7 of `bench.py --gpus 8` generate (frames r*1024 .. r*1024+1023), run on one GPU
one after the other: status, totals, and a sample of frames of each against the oracle (GPU box).
   python tools/rank_batches.py [first_rank] [last_rank]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from hope_amd import synth
from hope_amd.capi import HopeContext, HopeLib
from parity import assert_frame_equal

r0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
r1 = int(sys.argv[2]) if len(sys.argv) > 2 else 7
F, H, W = 1024, 480, 640
dev = torch.device("cuda", 0)
hip = HopeLib()
orc = HopeLib(os.path.join(ROOT, "oracle", "_build", "libhope_oracle.so"), "hope_oracle_")
ctx = HopeContext(hip)
for rank in range(r0, r1 + 1):
    depth = torch.empty((F, H, W), dtype=torch.uint16, device=dev)
    ts, qs = [], []
    for f0 in range(0, F, 32):
        d, t, q = synth.make_depth_frames(32, first_frame=rank * F + f0, device=dev)
        depth[f0:f0 + 32] = d
        ts.append(t)
        qs.append(q)
    t_np, q_np = torch.cat(ts).numpy(), torch.cat(qs).numpy()
    torch.cuda.synchronize()
    ctx.set_frame_poses(t_np, q_np)
    t0 = time.time()
    st = ctx.segment_depth_device(depth.data_ptr(), W, H, F)
    ms = 1e3 * (time.time() - t0)
    tot = [sum(getattr(ctx.counts(f), k) for f in range(F)) for k in ("n_raw", "n_vox", "n_fit", "n_planes")]
    print("rank", rank, "status", st, "ms", round(ms, 1), "raw/vox/fit/planes", tot, flush=True)
    sample = [0, 137, 511, 777, 1023]
    sub = np.stack([depth[f].cpu().numpy().view(np.uint16) for f in sample])
    o = HopeContext(orc)
    o.set_frame_poses(t_np[sample], q_np[sample])
    o.segment_depth(sub)
    g = HopeContext(hip)
    g.set_frame_poses(t_np[sample], q_np[sample])
    g.segment_depth(sub)
    for k, f in enumerate(sample):
        assert_frame_equal(o, g, k)
        a, b = ctx.counts(f), g.counts(k)
        assert (a.n_raw, a.n_vox, a.n_fit, a.n_clusters, a.n_planes) == (b.n_raw, b.n_vox, b.n_fit, b.n_clusters, b.n_planes)
    print("rank", rank, "sampled frames equal", flush=True)
print("done")
