"""world_size-2 gloo test of the frame sharding + rank-0 gather (the N>1 path of bench.py and of
hope_amd.sharding).  Runs on the CPU with the oracle standing in for the GPU library: what is under
test is the partition / gather logic, not the kernels."""
import os
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLDEN, ORACLE_LIB, ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hope_amd.capi import HopeContext, HopeLib
    from hope_amd.sharding import gather_results, segment_depth_sharded
    z = np.load(os.path.join(GOLDEN, "synth_frames_0_2.npz"))
    ctx = HopeContext(HopeLib(ORACLE_LIB, "hope_oracle_"))
    lo, hi, local = segment_depth_sharded(ctx, z["depth"], z["t"], z["q"], rank, world)
    merged = gather_results(local, dist, dst=0)
    dist.barrier()
    if rank == 0:
        q.put([(r["frame"], r["n_raw"], r["n_vox"], r["n_fit"], r["n_planes"],
                np.asarray(r["clusters"]).tobytes()) for r in merged])
    dist.destroy_process_group()


def test_frame_partition():
    from hope_amd.sharding import frame_partition
    for n in (0, 1, 3, 7, 1024):
        for world in (1, 2, 3, 8):
            spans = [frame_partition(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_match_one(oracle_lib):
    from hope_amd.capi import HopeContext
    z = np.load(os.path.join(GOLDEN, "synth_frames_0_2.npz"))
    ctx = HopeContext(oracle_lib)
    ctx.set_frame_poses(z["t"], z["q"])
    ctx.segment_depth(z["depth"])
    single = [(f, int(ctx.counts(f).n_raw), int(ctx.counts(f).n_vox), int(ctx.counts(f).n_fit),
               int(ctx.counts(f).n_planes), ctx.clusters(f).tobytes()) for f in range(3)]
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [mpctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    merged = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert merged == single


# ---- one cloud across two ranks --------------------------------------------
def _cloud(n=60000):
    from hope_amd import synth
    xyz = synth.make_multilevel_cloud(n, first_index=4321, device="cpu").numpy()
    xyz[::997, 1] = np.nan
    return xyz


def _cloud_worker(rank, world, port, q, apply):
    import torch
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hope_amd.capi import HopeContext, HopeLib
    from hope_amd.sharding import frame_partition, segment_cloud_sharded
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from parity import summary
    xyz = _cloud()
    if apply:
        xyz = xyz * np.float32(0.25) + np.array([0, 0, 0.2], np.float32)
    lo, hi = frame_partition(xyz.shape[0], world, rank)
    ctx = HopeContext(HopeLib(ORACLE_LIB, "hope_oracle_"))
    if apply:
        ctx.set_pose(np.array([0.1, -0.2, 0.3], np.float32), np.array([0.02, -0.01, 0.38, 0.92], np.float32))
    info = segment_cloud_sharded(ctx, torch.from_numpy(xyz[lo:hi].copy()), dist, apply_filter_transform=apply)
    dist.barrier()
    q.put((rank, summary(ctx, 0), info["sent"], info["received"]))
    dist.destroy_process_group()


def test_band_partition():
    from hope_amd.sharding import band_partition
    hist = np.array([0, 5, 0, 100, 3, 0, 0, 40, 40, 1], np.int64)
    for world in (1, 2, 3, 8):
        own = band_partition(hist, world)
        assert own.min() >= 0 and own.max() < world and np.all(np.diff(own) >= 0)
    assert band_partition(np.zeros(4, np.int64), 4).tolist() == [0, 0, 0, 0]


def test_one_cloud_on_two_ranks_matches_one_process(oracle_lib):
    from hope_amd.capi import HopeContext
    from parity import summary
    for apply in (False, True):
        xyz = _cloud()
        ctx = HopeContext(oracle_lib)
        if apply:
            xyz = xyz * np.float32(0.25) + np.array([0, 0, 0.2], np.float32)
            ctx.set_pose(np.array([0.1, -0.2, 0.3], np.float32), np.array([0.02, -0.01, 0.38, 0.92], np.float32))
        ctx.segment_cloud(xyz, apply_filter_transform=apply)
        single = summary(ctx, 0)
        assert single["n_planes"] > 0
        mpctx = mp.get_context("spawn")
        q = mpctx.Queue()
        port = 31500 + (os.getpid() % 2000) + (7 if apply else 0)
        procs = [mpctx.Process(target=_cloud_worker, args=(r, 2, port, q, apply)) for r in range(2)]
        for p in procs:
            p.start()
        got = [q.get(timeout=180) for _ in range(2)]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        for rank, summ, sent, received in got:
            assert summ == single, rank                      # every rank ends with the whole-cloud result
            assert sum(sent) > 0 and min(received) > 0       # both bands are populated: points really moved
