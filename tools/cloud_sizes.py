"""Largest cluster sizes of the config-4 scene (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hope_amd import synth
from hope_amd.capi import HopeContext, HopeLib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
xyz = torch.cat([synth.make_multilevel_cloud(min(10_000_000, n - i0), first_index=i0, device=dev) for i0 in range(0, n, 10_000_000)])
ctx = HopeContext(HopeLib())
torch.cuda.synchronize()
st = ctx.segment_cloud_device(xyz.data_ptr(), n, 12)
c = ctx.counts(0)
print("status", st, c.n_raw, c.n_vox, c.n_fit, c.n_clusters)
cl = ctx.clusters(0)
sz = np.sort(cl["size"])[::-1]
print("clusters", len(sz), "largest", sz[:12].tolist(), "sum", int(sz.sum()))
