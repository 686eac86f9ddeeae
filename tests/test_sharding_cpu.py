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
