import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hope_amd import synth
from hope_amd.capi import HopeContext, HopeLib, HopeConfig
F, W, H = 1024, 640, 480
dev = torch.device("cuda:0")
ds, ts, qs = [], [], []
for f0 in range(0, F, 128):
    d, t, q = synth.make_depth_frames(min(128, F - f0), first_frame=f0, device=dev)
    ds.append(d); ts.append(t.cpu()); qs.append(q.cpu())
depth = torch.cat(ds)
lib = HopeLib()
def run(ctx, n, tag):
    ctx.set_frame_poses(torch.cat(ts).numpy(), torch.cat(qs).numpy())
    for i in range(n):
        t0 = time.perf_counter()
        st = ctx.segment_depth_device(depth.data_ptr(), W, H, F)
        print(tag, i, "%.3f ms" % ((time.perf_counter() - t0) * 1e3), "status", st, ctx.last_error() if st < 0 else "", flush=True)
order = sys.argv[1]
mk = {"1": lambda: HopeContext(lib), "2": lambda: HopeContext(lib, sub_batches=2)}
keep = []
for ch in order:
    c = mk[ch](); keep.append(c)
    run(c, 6, "sb" + ch)
