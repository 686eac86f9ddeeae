"""One cloud across two ranks with the HIP library on both (the two processes share cuda:0; the exchanges
go over gloo with host tensors): the whole multi-rank orchestration of hope_amd.sharding -- bounding-box and
histogram all-reduces, the all-to-all by z band, the gathered VoxelGrid output, A6 and A8 shared by band
through hope_segment_downsampled_staged -- must end, on every rank, with the oracle's single-process result."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from test_sharding_cpu import _cloud


def _gpu_cloud_worker(rank, world, port, q):
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hope_amd.capi import HopeContext, HopeLib
    from hope_amd.sharding import frame_partition, segment_cloud_sharded
    from parity import summary
    xyz = _cloud(120000)
    lo, hi = frame_partition(xyz.shape[0], world, rank)
    ctx = HopeContext(HopeLib())
    info = segment_cloud_sharded(ctx, torch.from_numpy(xyz[lo:hi].copy()), dist)
    dist.barrier()
    q.put((rank, summary(ctx, 0), info["own_vox"], info["own_fit"]))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_one_cloud_on_two_ranks_hip(oracle_lib):
    from hope_amd.capi import HopeContext
    from parity import summary
    xyz = _cloud(120000)
    ref = HopeContext(oracle_lib)
    ref.segment_cloud(xyz)
    single = summary(ref, 0)
    assert single["n_planes"] > 0
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [mpctx.Process(target=_gpu_cloud_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, summ, own_vox, own_fit in got:
        assert summ == single, rank
        assert min(own_vox) > 0 and sum(own_fit) == single["n_fit"]   # both bands did part of A6 and A8
