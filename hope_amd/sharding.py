"""Multi-GPU use of the path: frames are independent extraction problems, so a batch is sharded
by contiguous frame range (one process per GPU) and only the small plane records are gathered on
rank 0.  There is no data-path collective ("Frames: yes, trivially").
A single huge cloud (configs 3/4) is not sharded yet: see DESIGN.md "Multi-GPU"."""
import numpy as np


def frame_partition(n_frames, world, rank):
    """Contiguous, balanced [lo, hi) frame range of `rank` (first n_frames % world ranks get one more)."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def segment_depth_sharded(ctx, depth, t, q, rank, world):
    """Runs this rank's share of a [n, h, w] uint16 batch through `ctx` (any implementation of the
    C ABI) and returns (lo, hi, [per-frame dict])."""
    n = depth.shape[0]
    lo, hi = frame_partition(n, world, rank)
    out = []
    if hi > lo:
        ctx.set_frame_poses(t[lo:hi], q[lo:hi])
        ctx.segment_depth(depth[lo:hi])
        for f in range(hi - lo):
            c = ctx.counts(f)
            out.append({"frame": lo + f, "n_raw": int(c.n_raw), "n_vox": int(c.n_vox), "n_fit": int(c.n_fit),
                        "n_planes": int(c.n_planes), "clusters": ctx.clusters(f).copy()})
    return lo, hi, out


def gather_results(local, dist, dst=0):
    """Gathers the per-frame records of every rank on `dst`, ordered by frame."""
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        parts = [None] * world
        dist.gather_object(local, parts, dst=dst)
        merged = [r for p in parts for r in p]
        assert [r["frame"] for r in merged] == sorted(r["frame"] for r in merged)
        return merged
    dist.gather_object(local, None, dst=dst)
    return None


def records_equal(a, b):
    return (a["frame"] == b["frame"] and a["n_raw"] == b["n_raw"] and a["n_vox"] == b["n_vox"] and
            a["n_fit"] == b["n_fit"] and a["n_planes"] == b["n_planes"] and
            np.asarray(a["clusters"]).tobytes() == np.asarray(b["clusters"]).tobytes())
