"""One cloud across two ranks with the HIP library on both (the two processes share cuda:0; the exchanges
go over gloo with host tensors): the whole multi-rank orchestration of hope_amd.sharding -- bounding-box and
histogram all-reduces, the all-to-all by z band, the gathered VoxelGrid output, A6 and A8 shared by band
through hope_segment_downsampled_staged, and A9..A12 either per z band (bands cut at empty layers, records
merged on rank 0) or replicated (edges cross the cuts) -- must end with the oracle's single-process result."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from test_sharding_cpu import _cloud, _layer_cloud, _ramp_cloud, _single_result

SCENES = {"multilevel": lambda: _cloud(120000), "layers": _layer_cloud, "ramp": _ramp_cloud}


def _gpu_cloud_worker(rank, world, port, q, scene):
    import torch
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from hope_amd.capi import HopeContext, HopeLib
    from hope_amd.sharding import frame_partition, segment_cloud_sharded
    from parity import summary
    xyz = SCENES[scene]()
    lo, hi = frame_partition(xyz.shape[0], world, rank)
    ctx = HopeContext(HopeLib())
    info = segment_cloud_sharded(ctx, torch.from_numpy(xyz[lo:hi].copy()), dist)
    dist.barrier()
    res = info["result"]
    if res is not None:
        res = {"clusters": np.asarray(res["clusters"]).tobytes(), "labels": np.asarray(res["labels"]).tobytes(),
               "n_segments": res["n_segments"], "n_clusters": res["n_clusters"], "n_planes": res["n_planes"]}
    q.put((rank, summary(ctx, 0), info["own_vox"], info["own_fit"], info["path"], res))
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("scene,path", [("multilevel", None), ("layers", "z-bands"), ("ramp", "replicated")])
def test_one_cloud_on_two_ranks_hip(oracle_lib, scene, path):
    from hope_amd.capi import HopeContext
    from parity import summary
    xyz = SCENES[scene]()
    ref = HopeContext(oracle_lib)
    ref.segment_cloud(xyz)
    single = summary(ref, 0)
    want = _single_result(oracle_lib, xyz)
    assert single["n_clusters"] > 0
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = 33500 + (os.getpid() % 2000) + len(scene)
    procs = [mpctx.Process(target=_gpu_cloud_worker, args=(r, 2, port, q, scene)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=240) for _ in range(2)], key=lambda g: g[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, summ, own_vox, own_fit, got_path, res in got:
        for k in ("n_raw", "n_vox", "n_fit", "grid", "min_b", "sha_downsampled", "sha_normals", "sha_fit_indices"):
            assert summ[k] == single[k], (rank, k)                        # A1..A7 of the whole cloud on every rank
        assert sum(own_fit) == single["n_fit"]
        if path is not None:
            assert got_path == path
        if got_path == "replicated":
            assert summ == single, rank
    assert got[0][5] == want                                               # the merged records and labels, rank 0
    if scene != "ramp":
        assert min(got[0][2]) > 0                                          # both bands did part of A6 and A8


def _bench_line(extra, env_extra):
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--frames", "8",
           "--points", "400000", "--no-cpu-baseline", "--no-stream"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_starts_its_own_ranks():
    """bench.py --gpus 2 without a launcher: the parent starts two ranks (here both on device 0, collectives over gloo:
    RCCL refuses two ranks on one device), rank 0 prints one line with the frames headline and the sharded cloud."""
    one = _bench_line(["--gpus", "1"], {})
    two = _bench_line(["--gpus", "2"], {"HOPE_BENCH_RANKS_ON_GPU0": "1", "HOPE_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "weak" and two["cloud"]["scaling"] == "strong"
    assert two["cloud"]["exchange"]["path"] in ("z-bands", "replicated")
    assert two["cloud"]["exchange"]["ranks_seen"] == 2
    assert two["cloud"]["planes"] == one["cloud"]["planes"] > 0
    assert two["cloud"]["points_downsampled"] == one["cloud"]["points_downsampled"]
    assert two["points_raw_per_step"] > one["points_raw_per_step"]       # two ranks, 8 frames each
    for line in (one, two):
        for key in ("roofline", "stages", "ms_per_step", "value"):
            assert key in line and key in line["cloud"]
