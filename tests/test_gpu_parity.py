"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same inputs.  Bar: bit-exact on every stage (stricter than north_star's 1e-5 relative for
coordinates and 1 mm for plane heights), and equal to the committed golden digests."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, synth_frames
from parity import assert_frame_equal, summary

from hope_amd.capi import HOPE_MODE_RT, HopeContext

pytestmark = pytest.mark.gpu


def golden():
    return json.load(open(os.path.join(GOLDEN, "oracle_golden.json")))


def both(hip_lib, oracle_lib, **kw):
    return HopeContext(hip_lib, **kw), HopeContext(oracle_lib, **kw)


def test_tum_example_pair(hip_lib, oracle_lib, tum_example):
    """BASELINE.json configs[0]: the reference's own CPU-runnable case."""
    depth, t, q = tum_example
    g, o = both(hip_lib, oracle_lib)
    for c in (g, o):
        c.set_pose(t, q)
        assert c.segment_depth(depth) == 0
    assert_frame_equal(o, g, 0, check_transformed=True)
    assert summary(g, 0) == golden()["tum_example"]
    assert g.counts(0).n_raw == 210004 and g.counts(0).n_vox == 5699 and g.counts(0).n_planes == 13


def test_synthetic_frame_batch(hip_lib, oracle_lib):
    """A batch with per-frame poses; frames of very different extent share one launch."""
    z = np.load(os.path.join(GOLDEN, "synth_frames_0_2.npz"))
    g, o = both(hip_lib, oracle_lib)
    for c in (g, o):
        c.set_frame_poses(z["t"], z["q"])
        assert c.segment_depth(z["depth"]) == 0
    for f in range(3):
        assert_frame_equal(o, g, f, check_transformed=(f == 1))
        assert summary(g, f) == golden()["synth_frames"][f]


def test_larger_batch_and_frame_independence(hip_lib, oracle_lib):
    depth, t, q = synth_frames(12, first=100)
    g, o = both(hip_lib, oracle_lib)
    for c in (g, o):
        c.set_frame_poses(t, q)
        assert c.segment_depth(depth) == 0
    for f in range(12):
        assert_frame_equal(o, g, f)
    # a frame's result does not depend on its batch neighbours
    batch7 = summary(g, 7)
    g.set_frame_poses(t[7:8], q[7:8])
    g.segment_depth(depth[7:8])
    assert summary(g, 0) == batch7


def test_device_resident_input(hip_lib, oracle_lib):
    import torch
    depth, t, q = synth_frames(2, first=7)
    g, o = both(hip_lib, oracle_lib)
    o.set_frame_poses(t, q)
    o.segment_depth(depth)
    dd = torch.from_numpy(depth.view(np.int16)).cuda()
    torch.cuda.synchronize()
    g.set_frame_poses(t, q)
    assert g.segment_depth_device(dd.data_ptr(), 640, 480, 2) == 0
    for f in range(2):
        assert_frame_equal(o, g, f)


def test_point_cloud_mode(hip_lib, oracle_lib):
    """POINT_CLOUD input (plane_segment.cpp:118-120), packed / PointXYZ / PointXYZRGB strides."""
    xyz = np.load(os.path.join(GOLDEN, "loft_120k.npz"))["xyz"]
    g, o = both(hip_lib, oracle_lib)
    o.segment_cloud(xyz)
    for width in (3, 4, 8):
        a = np.zeros((xyz.shape[0], width), np.float32)
        a[:, :3] = xyz
        a[:, 3:] = 7.0
        assert g.segment_cloud(a) == 0
        assert_frame_equal(o, g, 0, check_transformed=(width == 4))
    assert summary(g, 0) == golden()["loft_120k"]


def test_cloud_with_filter_and_pose(hip_lib, oracle_lib):
    """The RT callback path: PassThrough on sensor z, then the tf pose (plane_segment.cpp:791-795)."""
    from hope_amd import synth
    xyz = synth.make_multilevel_cloud(150000, first_index=0, device="cpu").numpy()
    xyz = xyz * np.float32(0.25) + np.array([0, 0, 0.2], np.float32)   # bring z into [0.2, 1.7]
    xyz[::1000, 0] = np.nan
    t = np.array([0.1, -0.2, 0.3], np.float32)
    q = np.array([0.02, -0.01, 0.38, 0.92], np.float32)                # not normalised on purpose
    for mode in (0, HOPE_MODE_RT):
        g, o = both(hip_lib, oracle_lib, mode=mode, min_height=0.4, max_height=1.2)
        for c in (g, o):
            c.set_pose(t, q)
            c.segment_cloud(xyz, apply_filter_transform=True)
        assert o.counts(0).n_raw < xyz.shape[0]
        assert_frame_equal(o, g, 0, check_transformed=True)


def test_rpy_pose(hip_lib, oracle_lib):
    """hope_set_pose_rpy: the roll/pitch/yaw doTransform of the SYN cloud callback (transform.cpp:108-130,
    plane_segment.cpp:617-621)."""
    from test_oracle import rpy_scene
    for rpy in ((0.3, -0.2, 0.5), (-1.1, 0.4, 2.9), (0.0, 0.0, 0.0)):
        xyz = rpy_scene(*rpy)
        g, o = both(hip_lib, oracle_lib)
        for c in (g, o):
            c.set_pose_rpy(*rpy)
            c.segment_cloud(xyz, apply_filter_transform=True)
        assert o.counts(0).n_planes >= 1
        assert_frame_equal(o, g, 0, check_transformed=True)


def test_multilevel_scene_cloud(hip_lib, oracle_lib):
    """Config-4 shaped input at a size the oracle finishes in seconds."""
    from hope_amd import synth
    xyz = synth.make_multilevel_cloud(400000, first_index=12345, device="cpu").numpy()
    g, o = both(hip_lib, oracle_lib)
    for c in (g, o):
        assert c.segment_cloud(xyz) == 0
    assert_frame_equal(o, g, 0)
    assert g.counts(0).n_clusters > 20


def test_one_huge_cluster(hip_lib, oracle_lib):
    """A floor of 8 m x 8 m: one cluster of more than 16384 members (dozens of tiles of k_cluster_verify_b's
    chains, the union-find tiles of several workgroups)."""
    rng = np.random.default_rng(5)
    n = 400000
    xyz = np.concatenate([rng.uniform(-4, 4, (n, 2)), rng.normal(0.0, 0.002, (n, 1))], 1).astype(np.float32)
    g, o = both(hip_lib, oracle_lib)
    for c in (g, o):
        assert c.segment_cloud(xyz) == 0
    assert_frame_equal(o, g, 0)
    assert g.clusters(0)["size"].max() > 16384
    # 14 m x 14 m: past kVBigFrom = 32768 members, where the launch with the longer tiles takes the cluster over
    n = 1200000
    xyz = np.concatenate([rng.uniform(-7, 7, (n, 2)), rng.normal(0.0, 0.002, (n, 1))], 1).astype(np.float32)
    for c in (g, o):
        assert c.segment_cloud(xyz) == 0
    assert_frame_equal(o, g, 0)
    assert g.clusters(0)["size"].max() > 40000 and g.counts(0).n_planes >= 1


def test_large_cluster_sums_by_integer_blocks(hip_lib, oracle_lib, monkeypatch, tum_example):
    """verify_big.hpp / exact_sum.hpp: every cluster of 64 members and more takes the path that the floors of
    configs 3 and 4 take (HOPE_DEBUG_VERIFY_BIG lowers its threshold from 32768), on depth frames, a cloud with
    walls and clutter and a floor; HOPE_DEBUG_VERIFY_BIG=0 keeps every cluster on the chain kernel."""
    from hope_amd import synth
    depth, t, q = synth_frames(6, first=40)
    cloud = synth.make_multilevel_cloud(400000, first_index=777, device="cpu").numpy()
    rng = np.random.default_rng(6)
    n = 500000
    floor = np.concatenate([rng.uniform(-5, 5, (n, 2)), rng.normal(0.0, 0.002, (n, 1))], 1).astype(np.float32)
    o = HopeContext(oracle_lib)
    for knob in ("64", "3", "0"):
        monkeypatch.setenv("HOPE_DEBUG_VERIFY_BIG", knob)
        g = HopeContext(hip_lib)
        monkeypatch.delenv("HOPE_DEBUG_VERIFY_BIG")
        for c in (g, o):
            c.set_frame_poses(t, q)
            assert c.segment_depth(depth) == 0
        for f in range(6):
            assert_frame_equal(o, g, f)
        st = g.xs_stats()
        assert (st["top_hits"] > 0) == (knob != "0"), st
        for xyz in (cloud, floor):
            for c in (g, o):
                assert c.segment_cloud(xyz) == 0
            assert_frame_equal(o, g, 0)
        d, tt, qq = tum_example
        for c in (g, o):
            c.set_pose(tt, qq)
            assert c.segment_depth(d) == 0
        assert_frame_equal(o, g, 0)


def test_cloud_voxel_stage_by_partition_or_sort(hip_lib, oracle_lib, monkeypatch):
    """A5 of a single cloud: stable partition passes + one wave per 512-cell bucket (voxel_part.hpp, the default)
    and the previous stage (keys, radix sort, run lengths, k_centroids_w: HOPE_DEBUG_VOXEL_PART=0), on clouds of
    one, two and three partition passes, with dropped points, and with one voxel that holds thousands of points."""
    from hope_amd import synth
    rng = np.random.default_rng(21)
    small = synth.make_multilevel_cloud(300000, first_index=5, device="cpu").numpy() * np.float32(0.08)  # 2.4 m: one pass
    mid = synth.make_multilevel_cloud(500000, first_index=99, device="cpu").numpy()                      # 30 m: two passes
    wide = np.concatenate([mid * np.float32(3.0), mid[:50000] * np.float32(8.0)])                        # 240 m: three
    dense = np.concatenate([mid[:200000], np.float32([3.0, 3.0, 1.0]) + rng.uniform(0, 0.019, (6000, 3)).astype(np.float32)])
    dense = dense[rng.permutation(len(dense))]
    holes = mid.copy()
    holes[::7, 2] = np.nan
    o = HopeContext(oracle_lib)
    for knob in (None, "0"):
        if knob is not None:
            monkeypatch.setenv("HOPE_DEBUG_VOXEL_PART", knob)
        g = HopeContext(hip_lib)
        monkeypatch.delenv("HOPE_DEBUG_VOXEL_PART", raising=False)
        for xyz in (small, mid, wide, dense, holes):
            for c in (g, o):
                assert c.segment_cloud(xyz) == 0
            assert_frame_equal(o, g, 0)
        t = np.array([0.1, -0.2, 0.3], np.float32)
        q = np.array([0.02, -0.01, 0.38, 0.92], np.float32)
        posed = mid * np.float32(0.25) + np.array([0, 0, 0.2], np.float32)
        for c in (g, o):
            c.set_pose(t, q)
            c.segment_cloud(posed, apply_filter_transform=True)
        assert o.counts(0).n_raw < posed.shape[0]
        assert_frame_equal(o, g, 0, check_transformed=True)
        for c in (g, o):
            c.set_pose(np.zeros(3, np.float32), np.array([0, 0, 0, 1], np.float32))


def test_rt_mode_and_height_window(hip_lib, oracle_lib):
    z = np.load(os.path.join(GOLDEN, "synth_frames_0_2.npz"))
    g, o = both(hip_lib, oracle_lib, mode=HOPE_MODE_RT, min_height=0.6, max_height=0.9)
    for lo, hi in ((0.6, 0.9), (-0.1, 0.3), (3.0, 4.0)):
        for c in (g, o):
            c.set_height_window(lo, hi)
            c.set_frame_poses(z["t"], z["q"])
            c.segment_depth(z["depth"])
        for f in range(3):
            assert_frame_equal(o, g, f)
    gold = golden()["rt_frame0"]
    g.set_height_window(0.6, 0.9)
    g.segment_depth(z["depth"])
    m = g.max_plane(0)
    assert {"valid": m.valid, "n_points": m.n_points, "z": float(m.z), "members": g.max_plane_members(0),
            "n_planes": g.counts(0).n_planes} == gold


def test_rt_winner_merges_more_than_sixteen_clusters(hip_lib, oracle_lib, tmp_path):
    """plane_segment.cpp:943-960: no limit on the clusters merged into the max plane.  24 patches at one
    height: the member list, the point count and the adapter's max_plane_cloud_ hold all of them."""
    import json
    import subprocess
    from conftest import patch_cloud
    xyz = patch_cloud(24)
    g, o = both(hip_lib, oracle_lib, mode=HOPE_MODE_RT, min_height=0.8, max_height=1.5, aggressive_merge=1)
    for c in (g, o):
        assert c.segment_cloud(xyz, apply_filter_transform=True) == 0
    assert_frame_equal(o, g, 0)
    m, mem = g.max_plane(0), g.max_plane_members(0)
    cl = g.clusters(0)
    assert m.n_members == len(mem) >= 20 and m.n_points == sum(int(cl["size"][i]) for i in mem)
    rt = os.path.join(ROOT, "tests", "_build", "rt_driver_hip")
    f = tmp_path / "patches.f32"
    xyz.tofile(f)
    r = subprocess.run([rt, str(f), "0", "0", "0", "0", "0", "0", "1", "0.8", "1.5", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["points"] == m.n_points   # max_plane_cloud_->size()


def test_edge_cases(hip_lib, oracle_lib):
    g, o = both(hip_lib, oracle_lib)
    cases_depth = [np.zeros((2, 480, 640), np.uint16), np.full((1, 480, 640), 40000, np.uint16)]
    mixed = synth_frames(2, first=3)[0].copy()
    mixed[1] = 0                                   # an empty frame inside a batch
    cases_depth.append(mixed)
    for d in cases_depth:
        sa, sb = o.segment_depth(d), g.segment_depth(d)
        assert sa == sb
        for f in range(d.shape[0]):
            assert_frame_equal(o, g, f)
    grid = np.stack(np.meshgrid(np.arange(2) * 0.05 + 0.02, np.arange(3) * 0.05 + 0.02), -1).reshape(-1, 2)
    tiny = np.concatenate([grid, np.full((6, 1), 0.5)], 1).astype(np.float32)
    clouds = [np.array([[0, 0, 0], [0.01, 0, 0]], np.float32), tiny, tiny[:1], np.zeros((0, 3), np.float32),
              np.full((5, 3), np.nan, np.float32)]
    for xyz in clouds:
        sa, sb = o.segment_cloud(xyz), g.segment_cloud(xyz)
        assert sa == sb, (sa, sb, xyz.shape)
        assert_frame_equal(o, g, 0)
    # non-default thresholds and k
    depth, t, q = synth_frames(1, first=11)
    for kw in (dict(th_xy=0.03, th_z=0.01), dict(knn=12), dict(knn=3, min_cluster=5), dict(th_xy=0.08, th_z=0.04)):
        g2, o2 = both(hip_lib, oracle_lib, **kw)
        for c in (g2, o2):
            c.set_frame_poses(t, q)
            c.segment_depth(depth)
        assert_frame_equal(o2, g2, 0)


def test_neighbour_list_overflow_branch(hip_lib, oracle_lib, monkeypatch):
    """k_normals keeps at most 64 neighbours in registers and falls back to repeated selection
    beyond that; the knob shrinks the list to 4 so that the rare branch runs on every point."""
    depth, t, q = synth_frames(1, first=5)
    monkeypatch.setenv("HOPE_DEBUG_SMALL_NB_CAP", "1")
    g = HopeContext(hip_lib)
    monkeypatch.delenv("HOPE_DEBUG_SMALL_NB_CAP")
    o = HopeContext(oracle_lib)
    for c in (g, o):
        c.set_frame_poses(t, q)
        c.segment_depth(depth)
    assert_frame_equal(o, g, 0)


def test_depth_frames_through_the_per_pixel_sort(hip_lib, oracle_lib, monkeypatch):
    """Depth frames normally take the (chunk, voxel)-group voxel stage; the knob sends them through
    the per-pixel sort that point clouds use, which must give the same bits."""
    depth, t, q = synth_frames(3, first=21)
    monkeypatch.setenv("HOPE_DEBUG_PIXEL_SORT", "1")
    g = HopeContext(hip_lib)
    monkeypatch.delenv("HOPE_DEBUG_PIXEL_SORT")
    g2, o = HopeContext(hip_lib), HopeContext(oracle_lib)
    for c in (g, g2, o):
        c.set_frame_poses(t, q)
        assert c.segment_depth(depth) == 0
    for f in range(3):
        assert_frame_equal(o, g, f)
        assert_frame_equal(o, g2, f)


def test_group_records_through_the_device_wide_sort(hip_lib, oracle_lib, monkeypatch):
    """The (chunk, voxel) group records are emitted frame by frame and normally sorted per frame with a
    segmented radix sort; the knob sends them through the device-wide sort (the path of frames too large
    for one workgroup), which must give the same bits."""
    depth, t, q = synth_frames(3, first=33)
    monkeypatch.setenv("HOPE_DEBUG_SEGSORT", "0")
    g = HopeContext(hip_lib)
    monkeypatch.delenv("HOPE_DEBUG_SEGSORT")
    o = HopeContext(oracle_lib)
    for c in (g, o):
        c.set_frame_poses(t, q)
        assert c.segment_depth(depth) == 0
    for f in range(3):
        assert_frame_equal(o, g, f)


def test_group_records_through_the_segmented_sort(hip_lib, oracle_lib, monkeypatch):
    """Depth batches normally rank their voxels with an LDS bitmap per frame (voxel_rank.hpp: no sort).  The knob
    sends the group records through the per-frame segmented radix sort + run-length encoding instead: the path
    of frames whose grid or voxel count does not fit in LDS.  Also: a grid too large for the bitmap (1 cm leaves)
    takes that path by itself."""
    depth, t, q = synth_frames(3, first=33)
    monkeypatch.setenv("HOPE_DEBUG_VOXEL_RANK", "0")
    g = HopeContext(hip_lib)
    monkeypatch.delenv("HOPE_DEBUG_VOXEL_RANK")
    o = HopeContext(oracle_lib)
    for c in (g, o):
        c.set_frame_poses(t, q)
        assert c.segment_depth(depth) == 0
    for f in range(3):
        assert_frame_equal(o, g, f)
    g, o = both(hip_lib, oracle_lib, th_xy=0.01, th_z=0.01)
    for c in (g, o):
        c.set_frame_poses(t[:1], q[:1])
        assert c.segment_depth(depth[:1]) == 0
    cells = np.prod(np.array(list(o.counts(0).grid), np.int64))
    assert cells > 160 * 1024 * 8, cells       # more cells than LDS has bits
    assert_frame_equal(o, g, 0)


def test_neighbour_tile_side(hip_lib, oracle_lib, monkeypatch):
    """The second implementation of A6 / A8 (HOPE_DEBUG_NEIGHBOURS=tiles: LDS copies of column tiles instead of the
    direct reads of the default kernels) gives a workgroup a 32x32-column super tile (sparse frames) or one 16x16 tile
    (dense clouds), picked per call from the points per super tile; the knob forces either.  Same bits."""
    monkeypatch.setenv("HOPE_DEBUG_NEIGHBOURS", "tiles")
    from hope_amd import synth
    depth, t, q = synth_frames(2, first=61)
    xyz = synth.make_multilevel_cloud(200000, first_index=99, device="cpu").numpy()
    o, oc = HopeContext(oracle_lib), HopeContext(oracle_lib)
    o.set_frame_poses(t, q)
    o.segment_depth(depth)
    oc.segment_cloud(xyz)
    for side in ("16", "32"):
        monkeypatch.setenv("HOPE_DEBUG_TILE_SIDE", side)
        g = HopeContext(hip_lib)
        monkeypatch.delenv("HOPE_DEBUG_TILE_SIDE")
        g.set_frame_poses(t, q)
        g.segment_depth(depth)
        for f in range(2):
            assert_frame_equal(o, g, f)
        g.segment_cloud(xyz)
        assert_frame_equal(oc, g, 0)


def test_neighbour_kernels_direct_or_tiles(hip_lib, oracle_lib, monkeypatch):
    """A6 and A8 by the default kernels (one thread per point, the column index read through the caches) and by the
    LDS tile kernels, each alone and both: 16 frames, a multi-level scene and the loft fixture, all == the oracle."""
    from hope_amd import synth
    depth, t, q = synth_frames(16, first=200)
    clouds = [synth.make_multilevel_cloud(400000, first_index=777, device="cpu").numpy(),
              np.load(os.path.join(GOLDEN, "loft_120k.npz"))["xyz"]]
    o = HopeContext(oracle_lib)
    o.set_frame_poses(t, q)
    o.segment_depth(depth)
    want_frames = [summary(o, f) for f in range(16)]
    want_clouds = []
    for xyz in clouds:
        o.segment_cloud(xyz)
        want_clouds.append(summary(o, 0))
    for mode in (None, "tiles", "tiles-normals", "tiles-knn"):
        if mode:
            monkeypatch.setenv("HOPE_DEBUG_NEIGHBOURS", mode)
        g = HopeContext(hip_lib)
        monkeypatch.delenv("HOPE_DEBUG_NEIGHBOURS", raising=False)
        g.set_frame_poses(t, q)
        g.segment_depth(depth)
        assert [summary(g, f) for f in range(16)] == want_frames, mode
        for xyz, want in zip(clouds, want_clouds):
            g.segment_cloud(xyz)
            assert summary(g, 0) == want, mode


def test_count_read_backs_by_mailbox_or_copy(hip_lib, oracle_lib, monkeypatch):
    """The counts that size the next launches reach the host through a polled mailbox in pinned host memory
    (k_publish_globals); the knob restores the D2H copy + stream synchronisation.  Same results either way, and a ctx
    of each kind can be used in turn."""
    depth, t, q = synth_frames(2, first=71)
    o = HopeContext(oracle_lib)
    o.set_frame_poses(t, q)
    o.segment_depth(depth)
    monkeypatch.setenv("HOPE_DEBUG_SYNC", "memcpy")
    g0 = HopeContext(hip_lib)
    monkeypatch.delenv("HOPE_DEBUG_SYNC")
    g1 = HopeContext(hip_lib)
    for rep in range(2):
        for g in (g0, g1):
            g.set_frame_poses(t, q)
            assert g.segment_depth(depth) == 0
            for f in range(2):
                assert_frame_equal(o, g, f)


def test_depth_frames_through_two_passes(hip_lib, oracle_lib, monkeypatch):
    """Depth frames normally take one pass over the batch (k_bbox_records: records with packed cell codes in
    sparse slots, keys made once the bounding box is known).  The knob forces the two-pass path
    (k_bbox_groups + k_group_records), which is also the fallback when a cell leaves the code range --
    3 mm leaves put the far floor more than 1023 cells from the camera, so that call falls back by itself."""
    depth, t, q = synth_frames(2, first=44)
    monkeypatch.setenv("HOPE_DEBUG_TWO_PASS", "1")
    g = HopeContext(hip_lib)
    monkeypatch.delenv("HOPE_DEBUG_TWO_PASS")
    o = HopeContext(oracle_lib)
    for c in (g, o):
        c.set_frame_poses(t, q)
        assert c.segment_depth(depth) == 0
    for f in range(2):
        assert_frame_equal(o, g, f)
    g2, o2 = both(hip_lib, oracle_lib, th_xy=0.003, th_z=0.01)
    for c in (g2, o2):
        c.set_frame_poses(t, q)
        c.segment_depth(depth)
    for f in range(2):
        assert_frame_equal(o2, g2, f)


def test_depth_group_records_dense_sparse_and_overflowing(hip_lib, oracle_lib, monkeypatch):
    """The group records of a depth batch lie densely per frame by default (k_bbox_records<.., DENSE>: LDS staging,
    one atomic per wave, sort words in k_voxel_rank).  HOPE_DEBUG_DENSE_RECORDS=0 keeps the sparse slots and
    k_group_keys; HOPE_DEBUG_CAP_MUL=1 leaves a frame's range room for one record per chunk, so every frame overflows
    it and the call falls back to the two passes with the bounding boxes it already has."""
    depth, t, q = synth_frames(3, first=61)
    o = HopeContext(oracle_lib)
    o.set_frame_poses(t, q)
    assert o.segment_depth(depth) == 0
    for knob, val in (("HOPE_DEBUG_DENSE_RECORDS", "0"), ("HOPE_DEBUG_CAP_MUL", "1"), ("HOPE_DEBUG_CAP_MUL", "16")):
        monkeypatch.setenv(knob, val)
        g = HopeContext(hip_lib)
        monkeypatch.delenv(knob)
        g.set_frame_poses(t, q)
        assert g.segment_depth(depth) == 0
        for f in range(3):
            assert_frame_equal(o, g, f)
        o1 = HopeContext(oracle_lib)                 # (a second call on the same context, other batch size)
        o1.set_frame_poses(t[:1], q[:1])
        g.set_frame_poses(t[:1], q[:1])
        assert o1.segment_depth(depth[:1]) == 0 and g.segment_depth(depth[:1]) == 0
        assert_frame_equal(o1, g, 0)


def test_member_lists_by_frame_or_by_sort(hip_lib, oracle_lib, monkeypatch):
    """The member lists of the clusters: batches of 64 frames and more hand every fit point its place frame by frame
    (k_members_by_frame, one wave per frame), fewer frames and single clouds sort (cluster, index) pairs.
    HOPE_DEBUG_MEMBERS_SORT forces either on frames and on a cloud."""
    from hope_amd import synth
    depth, t, q = synth_frames(5, first=23)
    cloud = synth.make_multilevel_cloud(300000, first_index=5, device="cpu").numpy()
    o = HopeContext(oracle_lib)
    for knob in ("0", "1"):
        monkeypatch.setenv("HOPE_DEBUG_MEMBERS_SORT", knob)
        g = HopeContext(hip_lib)
        monkeypatch.delenv("HOPE_DEBUG_MEMBERS_SORT")
        for c in (g, o):
            c.set_frame_poses(t, q)
            assert c.segment_depth(depth) == 0
        for f in range(5):
            assert_frame_equal(o, g, f)
        for c in (g, o):
            assert c.segment_cloud(cloud) == 0
        assert_frame_equal(o, g, 0)


def test_voxelgrid_pass_through_of_a_wide_cloud(hip_lib, oracle_lib):
    """PCL's VoxelGrid refuses a grid of more than INT_MAX cells ("Leaf size is too small for the input
    dataset") and copies its input through (voxel_grid.hpp applyFilter); A6..A12 then run on the raw
    points.  Table-sized patches scattered over 300 m x 300 m x 20 m: 6000 x 6000 x 1000 cells."""
    rng = np.random.default_rng(11)
    parts = []
    for k in range(24):
        c = rng.uniform([-150, -150, 0], [150, 150, 20])
        n = 500 if k % 3 else 1500   # (the dense patches exceed the 32-neighbour list: selection by rescan)
        p = np.empty((n, 3), np.float32)
        p[:, :2] = c[:2] + rng.uniform(-0.25, 0.25, (n, 2))
        p[:, 2] = c[2] + rng.normal(0, 0.002, n)
        if k % 5 == 4:   # a wall: fails the normal gate
            p[:, 0] = c[0] + rng.normal(0, 0.002, n)
            p[:, 2] = c[2] + rng.uniform(0, 0.5, n)
        parts.append(p)
    xyz = np.concatenate(parts)
    xyz = xyz[rng.permutation(len(xyz))]
    xyz[7] = [-0.0, 3.0, 1.0]
    xyz[100] = [np.nan, 0, 0]
    g, o = both(hip_lib, oracle_lib)
    for c in (g, o):
        assert c.segment_cloud(xyz) == 0
    k = g.counts(0)
    assert k.vox_passthrough == 1 and k.n_vox == k.n_raw == len(xyz) - 1 and list(k.grid) == [0, 0, 0]
    assert k.n_fit > 5000 and k.n_planes >= 10
    assert_frame_equal(o, g, 0)
    assert np.array_equal(g.downsampled(0).view(np.uint32), np.delete(xyz, 100, 0).view(np.uint32))
    # with the sensor-frame filter and the pose in front
    g, o = both(hip_lib, oracle_lib, min_depth=-1.0e3, max_depth=1.0e3)
    for c in (g, o):
        c.set_pose([1.0, -2.0, 0.5], [0.0, 0.0, 0.38268343, 0.92387953])
        assert c.segment_cloud(xyz, apply_filter_transform=True) == 0
    assert g.counts(0).vox_passthrough == 1
    assert_frame_equal(o, g, 0, check_transformed=True)


def test_voxelgrid_pass_through_in_a_depth_batch(hip_lib, oracle_lib):
    """Depth frames whose grid exceeds INT_MAX cells (th_z of 0.5 um: some 75 x 50 x 2000000) share a batch
    with a frame small enough to be down-sampled; the batch leaves the (chunk, voxel)-group path and
    runs again on the per-point voxel stage, where a pass-through frame's points are their own voxels."""
    depth, t, q = synth_frames(3, first=40)
    depth = np.ascontiguousarray(depth[:, 100:220, 200:360])
    depth[1] = 0
    depth[1, 50:58, 70:82] = 1500   # a fronto-parallel patch: a handful of cells
    kw = dict(th_xy=0.02, th_z=5.0e-7)
    g, o = both(hip_lib, oracle_lib, **kw)
    st = []
    for c in (g, o):
        c.set_frame_poses(t, q)
        st.append(c.segment_depth(depth))
    assert st[0] == st[1] >= 0   # (worst frame status: the patch of frame 1 may hold no level normal)
    passed = [g.counts(f).vox_passthrough for f in range(3)]
    assert passed[0] == 1 and passed[1] == 0   # (frame 2 goes by its own extent; the oracle decides alike below)
    assert g.counts(0).n_vox == g.counts(0).n_raw > 10000
    for f in range(3):
        assert_frame_equal(o, g, f, check_transformed=(f == 2))


def test_extent_beyond_any_column_index_is_refused(hip_lib):
    """More than 2^30 columns of th_xy: neither a VoxelGrid nor a column index; HOPE_E_RANGE with a message."""
    xyz = np.array([[0, 0, 0], [4.0e4, 4.0e4, 1.0], [1, 1, 1]], np.float32)
    g = HopeContext(hip_lib, th_xy=0.001, th_z=0.001)
    with pytest.raises(Exception, match="columns"):
        g.segment_cloud(xyz)


def test_column_sort_with_and_without_the_z_cell(hip_lib, oracle_lib, monkeypatch):
    """VoxelGrid output is sorted by column alone (the stable sort keeps the voxel order, ascending in z
    cell); the knob puts the z cell back into the key, as adopted and passed-through clouds have it.
    Either order of a column's entries must give the same bits."""
    depth, t, q = synth_frames(3, first=61)
    monkeypatch.setenv("HOPE_DEBUG_COLSORT_Z", "1")
    g = HopeContext(hip_lib)
    monkeypatch.delenv("HOPE_DEBUG_COLSORT_Z")
    g2, o = HopeContext(hip_lib), HopeContext(oracle_lib)
    for c in (g, g2, o):
        c.set_frame_poses(t, q)
        assert c.segment_depth(depth) == 0
    for f in range(3):
        assert_frame_equal(o, g, f)
        assert_frame_equal(o, g2, f)


def test_column_index_by_counting_sort_or_radix_sort(hip_lib, oracle_lib, monkeypatch):
    """Frames whose column table fits in LDS get their column index from a per-frame counting sort
    (k_column_index); the knob sends them through keys + radix sort + table + gather, which larger grids take by
    themselves.  A cloud with tall columns (the loft: walls) exercises the put-back-in-order step."""
    depth, t, q = synth_frames(3, first=61)
    xyz = np.load(os.path.join(GOLDEN, "loft_120k.npz"))["xyz"]
    monkeypatch.setenv("HOPE_DEBUG_COL_INDEX", "0")
    g = HopeContext(hip_lib)
    monkeypatch.delenv("HOPE_DEBUG_COL_INDEX")
    g2, o, oc = HopeContext(hip_lib), HopeContext(oracle_lib), HopeContext(oracle_lib)
    oc.segment_cloud(xyz)
    for c in (g, g2, o):
        c.set_frame_poses(t, q)
        assert c.segment_depth(depth) == 0
    for f in range(3):
        assert_frame_equal(o, g, f)
        assert_frame_equal(o, g2, f)
    for c in (g, g2):
        c.segment_cloud(xyz)
        assert_frame_equal(oc, c, 0)


def test_knn_leftovers_through_the_coarse_index(hip_lib, oracle_lib, monkeypatch):
    """k-NN queries that the LDS tile cannot settle go through a whole-frame ring expansion: over the
    fine column index when few, over the coarse index of the fit points when many.  The knob sets the
    limit to 0 so that every left-over query takes the coarse path; same bits either way."""
    from hope_amd import synth
    depth, t, q = synth_frames(2, first=70)
    xyz = synth.make_multilevel_cloud(300000, first_index=777, device="cpu").numpy()
    monkeypatch.setenv("HOPE_DEBUG_KNN_FINE_LIMIT", "0")
    g_coarse = HopeContext(hip_lib)                  # a thread per left-over query (k_knn_todo_threads)
    monkeypatch.setenv("HOPE_DEBUG_TODO_THREADS", "0")
    g_waves = HopeContext(hip_lib)                   # a wave per query (k_knn_todo over the coarse index)
    monkeypatch.delenv("HOPE_DEBUG_TODO_THREADS")
    monkeypatch.setenv("HOPE_DEBUG_KNN_FINE_LIMIT", "1000000000")
    g_fine = HopeContext(hip_lib)
    monkeypatch.delenv("HOPE_DEBUG_KNN_FINE_LIMIT")
    o = HopeContext(oracle_lib)
    gs = (g_coarse, g_waves, g_fine)
    for c in gs + (o,):
        c.set_frame_poses(t, q)
        assert c.segment_depth(depth) == 0
    for f in range(2):
        for g in gs:
            assert_frame_equal(o, g, f)
    for c in gs + (o,):
        assert c.segment_cloud(xyz) == 0
    for g in gs:
        assert_frame_equal(o, g, 0)


def test_row_accessors_into_device_memory(hip_lib, oracle_lib):
    """hope_get_normals_rows / hope_get_knn_rows write a row range to host memory or straight into device
    memory (what a rank contributes to the all-gathers of the staged multi-GPU call)."""
    import torch
    from hope_amd import synth
    xyz = synth.make_multilevel_cloud(150000, first_index=5, device="cpu").numpy()
    g, o = both(hip_lib, oracle_lib)
    for c in (g, o):
        assert c.segment_cloud(xyz) == 0
    n_vox, n_fit = int(g.counts(0).n_vox), int(g.counts(0).n_fit)
    a, b = n_vox // 3, n_vox - 7
    fa, fb = n_fit // 4, n_fit - 3
    ref_n, ref_k = o.normals(0), o.knn(0)
    assert g.normals_rows(a, b - a).tobytes() == ref_n[a:b].tobytes()
    assert np.array_equal(g.knn_rows(fa, fb - fa), ref_k[fa:fb])
    dn = torch.empty((b - a, 4), dtype=torch.float32, device="cuda")
    dk = torch.empty((fb - fa, g.cfg.knn), dtype=torch.int32, device="cuda")
    g.normals_rows(a, b - a, out_ptr=dn.data_ptr(), out_on_device=True)
    g.knn_rows(fa, fb - fa, out_ptr=dk.data_ptr(), out_on_device=True)
    torch.cuda.synchronize()
    assert dn.cpu().numpy().tobytes() == ref_n[a:b].tobytes()
    assert np.array_equal(dk.cpu().numpy(), ref_k[fa:fb])
    assert o.normals_rows(a, b - a).tobytes() == ref_n[a:b].tobytes()
    assert np.array_equal(o.knn_rows(fa, fb - fa), ref_k[fa:fb])


def test_staged_cloud_entry_points(hip_lib, oracle_lib):
    """The device stages behind hope_amd.sharding.segment_cloud_sharded, driven here for three
    'ranks' one after the other in one process (the exchanges become numpy concatenations): the
    result must be the oracle's single-call result, bit for bit, and every stage must agree with the
    oracle's own staged entry points."""
    from hope_amd import synth
    from hope_amd.sharding import band_partition, frame_partition
    xyz = synth.make_multilevel_cloud(200000, first_index=99, device="cpu").numpy()
    xyz = (xyz * np.float32(0.25) + np.array([0, 0, 0.2], np.float32)).astype(np.float32)
    xyz[::1501, 0] = np.nan
    t = np.array([0.1, -0.2, 0.3], np.float32)
    q = np.array([0.02, -0.01, 0.38, 0.92], np.float32)
    world = 3
    g, o = both(hip_lib, oracle_lib)
    ref = HopeContext(oracle_lib)
    for c in (g, o, ref):
        c.set_pose(t, q)
    ref.segment_cloud(xyz, apply_filter_transform=True)
    for c in (g, o):
        shards = [np.ascontiguousarray(xyz[slice(*frame_partition(xyz.shape[0], world, r))]) for r in range(world)]
        boxes = [c.shard_bbox(s.ctypes.data, s.shape[0], 12, True, False) for s in shards]
        gb = np.concatenate([np.min([b[0][:3] for b in boxes], 0), np.max([b[0][3:] for b in boxes], 0)])
        n_raw = sum(b[1] for b in boxes)
        assert n_raw == ref.counts(0).n_raw
        mn, dv, _ = c.grid_from_bbox(gb)
        assert list(dv) == list(ref.counts(0).grid) and list(mn) == list(ref.counts(0).min_b)
        hist = sum(c.shard_z_histogram(s.ctypes.data, s.shape[0], 12, True, False, gb, int(dv[2])) for s in shards)
        assert hist.sum() == n_raw
        owner = band_partition(hist, world)
        routed, counts = [], []
        for s, b in zip(shards, boxes):
            out = np.zeros((max(b[1], 1), 3), np.float32)
            counts.append(c.shard_route(s.ctypes.data, s.shape[0], 12, True, False, gb, owner, world,
                                        out.ctypes.data, False))
            routed.append(out)
        vox = []
        for r in range(world):
            pieces = []
            for src in range(world):
                off = int(counts[src][:r].sum())
                pieces.append(routed[src][off:off + int(counts[src][r])])
            mine = np.ascontiguousarray(np.concatenate(pieces))
            c.downsample_cloud_bbox(mine.ctypes.data, mine.shape[0], 12, False, gb)
            vox.append(c.downsampled(0).copy())
        vox_all = np.ascontiguousarray(np.concatenate(vox))
        assert vox_all.tobytes() == ref.downsampled(0).tobytes()
        c.segment_downsampled(vox_all.ctypes.data, vox_all.shape[0], 12, False, gb, n_raw)
        assert summary(c, 0) == summary(ref, 0)
        # the same last step with A6 and A8 shared by z band (hope_segment_downsampled_staged)
        n_vox = vox_all.shape[0]
        args = (vox_all.ctypes.data, n_vox, 12, False, gb, n_raw)
        firsts = np.concatenate([[0], np.cumsum([v.shape[0] for v in vox])]).astype(np.int64)
        nrm = []
        for r in range(world):
            c.segment_downsampled_staged(*args, firsts[r], firsts[r + 1] - firsts[r], stop_after=1)
            nrm.append(c.normals(0)[firsts[r]:firsts[r + 1]].copy())
        normals_all = np.ascontiguousarray(np.concatenate(nrm))
        assert normals_all.tobytes() == ref.normals(0).tobytes()
        rows = []
        for r in range(world):
            c.segment_downsampled_staged(*args, firsts[r], firsts[r + 1] - firsts[r],
                                         normals_ptr=normals_all.ctypes.data, stop_after=2)
            fit_idx = c.fit_indices(0)
            lo, hi = np.searchsorted(fit_idx, [firsts[r], firsts[r + 1]])
            rows.append(c.knn(0)[lo:hi].copy())
        knn_all = np.ascontiguousarray(np.concatenate(rows))
        assert np.array_equal(knn_all, ref.knn(0))
        c.segment_downsampled_staged(*args, 0, n_vox, normals_ptr=normals_all.ctypes.data, knn_ptr=knn_all.ctypes.data)
        assert summary(c, 0) == summary(ref, 0)
    # (n_in differs by construction: the last stage's input is the down-sampled cloud)
    assert np.array_equal(ref.labels(0), g.labels(0)) and np.array_equal(ref.knn(0), g.knn(0))
    assert ref.clusters(0).tobytes() == g.clusters(0).tobytes()


def test_cloud_normals_over_the_occupancy_index_or_the_column_index(hip_lib, oracle_lib, monkeypatch):
    """A6 of a single cloud over the 3-D occupancy index (grid_index.hpp, the default) and over the column index
    (HOPE_DEBUG_GRID_INDEX=0); with neighbour lists longer than the ordering networks (cap of 2: one scan per pick);
    and on an adopted cloud that is NO VoxelGrid output (several points per cell, input order): the library notices
    that the derived cells do not ascend and takes the column index for that call."""
    from hope_amd import synth
    mid = synth.make_multilevel_cloud(400000, first_index=31, device="cpu").numpy()
    loft = np.load(os.path.join(GOLDEN, "loft_120k.npz"))["xyz"]
    o = HopeContext(oracle_lib)
    for env in ({}, {"HOPE_DEBUG_GRID_INDEX": "0"}, {"HOPE_DEBUG_SMALL_NB_CAP": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        g = HopeContext(hip_lib)
        for k in env:
            monkeypatch.delenv(k)
        for xyz in (mid, loft):
            for c in (g, o):
                assert c.segment_cloud(xyz) == 0
            assert_frame_equal(o, g, 0)
        # adopted as it is: a raw cloud, many points per cell, in input order
        raw = np.ascontiguousarray(mid[:60000] * np.float32(0.2))
        gb = np.concatenate([raw.min(0), raw.max(0)]).astype(np.float32)
        for c in (g, o):
            c.segment_downsampled(raw.ctypes.data, raw.shape[0], 12, False, gb, raw.shape[0])
        assert summary(g, 0) == summary(o, 0)
        assert np.array_equal(o.normals(0), g.normals(0), equal_nan=True) and np.array_equal(o.knn(0), g.knn(0))


def test_division_by_camera_constants(hip_lib, oracle_lib, monkeypatch):
    """A1's divisions by depth_scale / fx / fy run as a 3-instruction sequence once the library has
    checked it against the IEEE quotient for every possible dividend; the knob forces IEEE divisions.
    Both must give the oracle's bits, also for intrinsics other than TUM's."""
    depth, t, q = synth_frames(2, first=60)
    monkeypatch.setenv("HOPE_DEBUG_IEEE_DIV", "1")
    g_ieee = HopeContext(hip_lib)
    monkeypatch.delenv("HOPE_DEBUG_IEEE_DIV")
    g_fast, o = HopeContext(hip_lib), HopeContext(oracle_lib)
    for c in (g_ieee, g_fast, o):
        c.set_frame_poses(t, q)
        assert c.segment_depth(depth) == 0
    assert g_ieee.fast_division_active() == 0
    assert g_fast.fast_division_active() == 1      # TUM intrinsics pass the exhaustive check
    for f in range(2):
        assert_frame_equal(o, g_ieee, f, check_transformed=True)
        assert_frame_equal(o, g_fast, f, check_transformed=True)
    for kw in (dict(fx=525.0, fy=525.5, cx=319.5, cy=239.5, depth_scale=1000.0), dict(fx=450.3, fy=451.7, cx=300.1, cy=250.9, depth_scale=4999.0)):
        g2, o2 = both(hip_lib, oracle_lib, **kw)
        for c in (g2, o2):
            c.set_frame_poses(t, q)
            c.segment_depth(depth)
        for f in range(2):
            assert_frame_equal(o2, g2, f, check_transformed=True)


def test_odd_image_shapes(hip_lib, oracle_lib):
    """Widths that are not a multiple of the 64-pixel chunk, a ragged last chunk, tiny images."""
    depth, t, q = synth_frames(2, first=40)
    for (h, w) in ((37, 100), (480, 333), (1, 64), (3, 5), (65, 63)):
        d = np.ascontiguousarray(depth[:, 100:100 + h, 200:200 + w])
        g, o = both(hip_lib, oracle_lib)
        for c in (g, o):
            c.set_frame_poses(t, q)
            c.segment_depth(d)
        for f in range(2):
            assert_frame_equal(o, g, f, check_transformed=(h == 37))


def test_invalid_arguments(hip_lib):
    import ctypes as C
    g = HopeContext(hip_lib)
    assert hip_lib.fn["segment_depth_u16"](g._h, None, 640, 480, 1, 0) < 0
    assert hip_lib.fn["segment_cloud_xyz"](g._h, 1, 10, 10, 0, 0) < 0      # stride not a multiple of 4
    k = C.c_int(0)
    g.segment_cloud(np.zeros((4, 3), np.float32))
    from hope_amd.capi import HopeCounts
    cnt = HopeCounts()
    assert hip_lib.fn["frame_counts"](g._h, 5, C.byref(cnt)) < 0
    assert b"range" in hip_lib.fn["last_error"](g._h)
    # the round-2 entry points refuse what they cannot serve, on both libraries alike
    n64 = C.c_int64(0)
    pos = (C.c_int32 * 4)()
    assert hip_lib.fn["set_pose_rpy"](None, 0.0, 0.0, 0.0) < 0
    assert hip_lib.fn["get_cluster_hull_candidates"](g._h, 0, 0, pos, 4) < 0          # the 4-point cloud has no cluster
    assert hip_lib.fn["get_cluster_hull_candidates"](g._h, 3, 0, pos, 4) < 0          # no such frame
    assert hip_lib.fn["count_band_crossings"](g._h, 0, 0, 1, C.byref(n64)) < 0        # more fit points than there are
    assert hip_lib.fn["count_band_crossings"](g._h, 0, 0, 0, None) < 0
    assert hip_lib.fn["count_band_crossings"](g._h, 0, 0, 0, C.byref(n64)) == 0 and n64.value == 0


def test_sub_batches_on_their_own_streams(hip_lib, oracle_lib, monkeypatch):
    """With hope_config.sub_batches = n a batch of many frames runs as up to n frame ranges, each in a child context with
    its own stream, workspace and host thread (hope_hip.hip: segment_depth_split; off by default: measured slower
    than one range on MI355X).  Forced here on 14 frames (ranges of 3-4):
    every stage of every frame equals the oracle's, the per-frame accessors and the objects-on-plane step find
    their frame in the right range, and a later small call leaves the split state behind."""
    from test_upon_plane import winner_contour
    import ctypes
    depth, t, q = synth_frames(14, first=40)
    monkeypatch.setenv("HOPE_DEBUG_SUB_MIN", "3")
    g = HopeContext(hip_lib, mode=HOPE_MODE_RT, min_height=0.5, max_height=1.0, sub_batches=4)
    monkeypatch.delenv("HOPE_DEBUG_SUB_MIN")
    o = HopeContext(oracle_lib, mode=HOPE_MODE_RT, min_height=0.5, max_height=1.0)
    for c in (g, o):
        c.set_frame_poses(t, q)
        assert c.segment_depth(depth) == 0
        assert c.num_frames() == 14
    for f in range(14):
        assert_frame_equal(o, g, f, check_transformed=(f in (3, 4, 13)))
    host = ctypes.CDLL(os.path.join(ROOT, "hope_amd", "host", "libhope_host.so"))
    for f in (0, 6, 13):
        if not o.max_plane(f).valid:
            continue
        contour = winner_contour(o, host, f)
        na, ca = o.clusters_upon_plane(contour, frame=f)
        nb, cb = g.clusters_upon_plane(contour, frame=f)
        assert na == nb and len(ca) == len(cb)
        for (xa, ia), (xb, ib) in zip(ca, cb):
            assert np.array_equal(ia, ib) and xa.tobytes() == xb.tobytes()
    from hope_amd.capi import HopeCounts
    cnt = HopeCounts()
    assert hip_lib.fn["frame_counts"](g._h, 14, ctypes.byref(cnt)) < 0
    # the same pose for every frame (hope_set_pose) reaches the ranges too
    for c in (g, o):
        c.set_pose(t[5], q[5])
        c.segment_depth(depth)
    for f in (0, 5, 13):
        assert_frame_equal(o, g, f)
    # a small call afterwards is not split
    for c in (g, o):
        c.set_frame_poses(t[:2], q[:2])
        c.segment_depth(depth[:2])
        assert c.num_frames() == 2
    assert_frame_equal(o, g, 1)


def test_full_size_batch_properties(hip_lib, oracle_lib):
    """BASELINE.json configs[1] at full size (1024 frames generated on the device): the oracle cannot
    run all of it in seconds, so check size-independent properties plus exact parity on sampled frames."""
    import torch
    from hope_amd import synth
    F = 1024
    depth = torch.empty((F, 480, 640), dtype=torch.uint16, device="cuda")
    ts, qs = [], []
    for f0 in range(0, F, 64):
        d, t, q = synth.make_depth_frames(64, first_frame=f0, device="cuda")
        depth[f0:f0 + 64] = d
        ts.append(t)
        qs.append(q)
    t, q = torch.cat(ts).numpy(), torch.cat(qs).numpy()
    torch.cuda.synchronize()
    g = HopeContext(hip_lib)
    g.set_frame_poses(t, q)
    assert g.segment_depth_device(depth.data_ptr(), 640, 480, F) == 0
    valid = ((depth.to(torch.int32) > 1500) & (depth.to(torch.int32) < 40000)).sum(dim=(1, 2)).cpu().numpy()
    for f in range(F):
        c = g.counts(f)
        assert c.n_raw == valid[f]                      # 0.3 < d/5000 < 8.0  <=>  1500 < d < 40000
        assert 0 < c.n_fit <= c.n_vox <= c.n_raw
        assert c.n_planes <= c.n_clusters <= c.n_segments <= c.n_fit
    # 36 frames: every 31st plus a few more
    sample = sorted(set(list(range(0, F, 31)) + [255, 256, 511, 512, 767, 768, 1023]))
    digests = {f: summary(g, f) for f in sample}
    labels = {f: g.labels(f) for f in sample}
    for f in sample:
        cl = g.clusters(f)
        assert np.array_equal(np.bincount(labels[f][labels[f] >= 0], minlength=len(cl)), cl["size"])
        assert (cl["size"] >= 3).all() and np.all(np.diff(cl["seed_rank"]) > 0)
    # idempotence: a second pass over the same resident batch gives identical results
    assert g.segment_depth_device(depth.data_ptr(), 640, 480, F) == 0
    for f in sample:
        assert summary(g, f) == digests[f]
    # exact parity with the oracle on the sampled frames
    o = HopeContext(oracle_lib)
    for f in sample:
        o.set_pose(t[f], q[f])
        o.segment_depth(depth[f].cpu().numpy().view(np.uint16))
        assert summary(o, 0) == digests[f]


def _cloud_properties(g, n_in):
    c = g.counts(0)
    assert c.n_in == n_in and 0 < c.n_fit <= c.n_vox <= c.n_raw <= n_in
    assert c.n_planes <= c.n_clusters <= c.n_segments <= c.n_fit
    labels, cl = g.labels(0), g.clusters(0)
    assert len(cl) == c.n_clusters and int(cl["pass"].sum()) == c.n_planes
    assert np.array_equal(np.bincount(labels[labels >= 0], minlength=len(cl)), cl["size"])
    assert (cl["size"] >= 3).all() and np.all(np.diff(cl["seed_rank"]) > 0)
    fit = g.fit_indices(0)
    assert np.all(np.diff(fit) > 0) and fit[-1] < c.n_vox
    vox = g.downsampled(0)
    key = np.floor(vox * np.float32([20, 20, 50])).astype(np.int64) - np.array(list(c.min_b), np.int64)
    lin = key[:, 0] + key[:, 1] * c.grid[0] + key[:, 2] * c.grid[0] * c.grid[1]
    assert np.all(np.diff(lin) > 0)                      # VoxelGrid order: one centroid per cell, ascending index
    return summary(g, 0)


def test_full_size_loft_cloud(hip_lib, oracle_lib):
    """BASELINE.json configs[2] stand-in at full size (5 million points; loft.ply itself is not in the image):
    size-independent properties, idempotence, and exact parity with the oracle on the whole cloud."""
    from hope_amd import synth
    xyz = synth.make_loft_cloud(5_000_000)
    g, o = both(hip_lib, oracle_lib)
    assert g.segment_cloud(xyz) == 0
    d1 = _cloud_properties(g, xyz.shape[0])
    assert g.segment_cloud(xyz) == 0
    assert summary(g, 0) == d1
    assert o.segment_cloud(xyz) == 0
    assert_frame_equal(o, g, 0)


def test_full_size_multilevel_scene(hip_lib, oracle_lib):
    """BASELINE.json configs[3] at full size: the 10^8-point multi-level scene, generated on the device.  The
    oracle cannot run it in seconds: size-independent properties and idempotence on the whole scene, exact
    parity on a 6 m x 6 m crop of it (same density: ~5 million points, tens of points per voxel)."""
    import torch
    from hope_amd import synth
    n = 100_000_000
    parts = [synth.make_multilevel_cloud(10_000_000, first_index=i0, device="cuda") for i0 in range(0, n, 10_000_000)]
    xyz = torch.cat(parts)
    del parts
    torch.cuda.synchronize()
    g, o = both(hip_lib, oracle_lib)
    assert g.segment_cloud_device(xyz.data_ptr(), n, 12) == 0
    d1 = _cloud_properties(g, n)
    c = g.counts(0)
    assert c.n_raw == n and all(590 <= v <= 610 for v in list(c.grid)[:2])   # 30 m x 30 m at 0.05 m
    assert g.clusters(0)["size"].max() > 300_000                               # the floor: one huge cluster
    assert g.segment_cloud_device(xyz.data_ptr(), n, 12) == 0
    assert summary(g, 0) == d1
    crop = xyz[(xyz[:, 0] > 10.0) & (xyz[:, 0] < 16.0) & (xyz[:, 1] > 8.0) & (xyz[:, 1] < 14.0)].contiguous()
    torch.cuda.synchronize()
    del xyz
    assert 3_000_000 < crop.shape[0] < 8_000_000
    assert g.segment_cloud_device(crop.data_ptr(), crop.shape[0], 12) == 0
    assert o.segment_cloud(crop.cpu().numpy()) == 0
    assert_frame_equal(o, g, 0)
    assert g.counts(0).n_planes > 0


@pytest.mark.gpu
def test_hull_candidates_hold_every_hull_vertex(hip_lib, oracle_lib, tum_example):
    """F2: hope_get_cluster_hull_candidates keeps the members outside the octagon of the cluster's extreme points.
    Every vertex qhull (scipy; the library behind pcl::ConvexHull) finds on ALL points of a plane must be among
    them, the hull of the candidates alone has the same area, and on large planes they are a small fraction."""
    from scipy.spatial import ConvexHull
    from hope_amd import synth
    depth, t, q = tum_example
    g = HopeContext(hip_lib)
    g.set_pose(t, q)
    g.segment_depth(depth)
    cases = [(g, 0)]
    sd, st_, sq = synth_frames(3, first=5)
    g2 = HopeContext(hip_lib)
    g2.set_frame_poses(st_, sq)
    g2.segment_depth(sd)
    cases += [(g2, f) for f in range(3)]
    g3 = HopeContext(hip_lib)
    g3.segment_cloud(synth.make_multilevel_cloud(400000, first_index=0, device="cpu").numpy())
    cases.append((g3, 0))
    checked = big = 0
    for c, f in cases:
        vox, fit, cl = c.downsampled(f), c.fit_indices(f), c.clusters(f)
        for ci in range(len(cl)):
            size = int(cl["size"][ci])
            if not cl["pass"][ci] or size < 3:
                continue
            pts = vox[fit[c.cluster_indices(f, ci, size)]][:, :2].astype(np.float64)
            cand = c.cluster_hull_candidates(f, ci, size)
            assert np.all(np.diff(cand) > 0) and cand[0] >= 0 and cand[-1] < size
            if len(np.unique(pts, axis=0)) < 3 or np.linalg.matrix_rank(pts - pts[0]) < 2:
                continue
            full = ConvexHull(pts)
            assert set(full.vertices.tolist()) <= set(cand.tolist())
            sub = ConvexHull(pts[cand])
            assert abs(sub.volume - full.volume) <= 1e-12 * full.volume        # (2-D: volume is the area)
            checked += 1
            if size >= 4000:
                big += 1
                assert len(cand) < size // 3, (size, len(cand))
    assert checked >= 10 and big >= 1
    # the oracle side of the same entry point: the reference hulls every point
    o = HopeContext(oracle_lib)
    o.set_pose(t, q)
    o.segment_depth(depth)
    ocl = o.clusters(0)
    ci = int(np.argmax(ocl["size"]))
    assert np.array_equal(o.cluster_hull_candidates(0, ci, int(ocl["size"][ci])), np.arange(int(ocl["size"][ci])))


def test_hope_node_and_rt_adapter_on_the_device(oracle_lib, tum_example, tmp_path):
    """The C++ host layer linked on libhope_hip.so: `hope_node` (single pair, list mode) and the PlaneSegmentRT
    driver give the records the oracle gives through the same C ABI."""
    import json
    import subprocess
    from test_host_cpu import check_record, expected_planes, run_node, write_tum_dataset
    node = os.path.join(ROOT, "hope_amd", "host", "hope_node")
    rt = os.path.join(ROOT, "tests", "_build", "rt_driver_hip")
    assert os.path.exists(node) and os.path.exists(rt), "build hope_amd/host and tests/ before the GPU run"
    depth, t, q = tum_example
    d = write_tum_dataset(tmp_path, [depth], [t], [q])
    args = [str(d) + "/", "rgb/0.png", "depth/0.png"] + ["%.4f" % v for v in t] + ["%.4f" % v for v in q]
    r, recs = run_node(node, args, tmp_path)
    assert r.returncode == 0, r.stderr
    k, planes = expected_planes(oracle_lib, depth, np.float32([float("%.4f" % v) for v in t]), np.float32([float("%.4f" % v) for v in q]))
    check_record(recs[0], k, planes)
    g = np.load(os.path.join(GOLDEN, "synth_frames_0_2.npz"))
    d = write_tum_dataset(tmp_path / "list", g["depth"], g["t"], g["q"])
    r, recs = run_node(node, [d], tmp_path, hull=False)
    assert r.returncode == 0 and len(recs) == 3, r.stderr
    for i, rec in enumerate(recs):
        k, planes = expected_planes(oracle_lib, g["depth"][i], g["t"][2], g["q"][i])
        check_record(rec, k, planes)
    # F4 plane-ID tracking across the frames of the list (plane_segment.cpp:496-517, utilities.cpp:1063-1159): the
    # device build hands out the ids the oracle build of the same node hands out; first detection = 0..n-1
    node_oracle = os.path.join(ROOT, "tests", "_build", "hope_node_oracle")
    r2, recs2 = run_node(node_oracle, [d], tmp_path, hull=False)
    assert r2.returncode == 0 and len(recs2) == 3, r2.stderr
    assert [p["id"] for p in recs[0]["planes"]] == list(range(len(recs[0]["planes"])))
    for a, b in zip(recs, recs2):
        assert [p["id"] for p in a["planes"]] == [p["id"] for p in b["planes"]]
        assert len(set(p["id"] for p in a["planes"])) == len(a["planes"])
    # RT adapter: the same driver on the oracle build and on the device build prints the same line
    z = depth.astype(np.float32) / np.float32(5000.0)
    cols = (np.arange(640, dtype=np.float32) - np.float32(318.6))[None, :]
    rows = (np.arange(480, dtype=np.float32) - np.float32(255.3))[:, None]
    cloud = np.stack([cols * z / np.float32(517.0), rows * z / np.float32(517.0), z], -1).reshape(-1, 3)
    f = tmp_path / "cloud.f32"
    cloud.astype(np.float32).tofile(f)
    rt_oracle = os.path.join(ROOT, "tests", "_build", "rt_driver_oracle")
    for lo, hi, aggr in ((0.5, 1.5, 1), (-0.2, 0.3, 0)):
        a = [str(f)] + [repr(float(v)) for v in t] + [repr(float(v)) for v in q] + [str(lo), str(hi), str(aggr)]
        out = [subprocess.run([b] + a, capture_output=True, text=True, timeout=300) for b in (rt, rt_oracle)]
        assert out[0].returncode == 0 and out[1].returncode == 0, (out[0].stderr, out[1].stderr)
        assert json.loads(out[0].stdout.splitlines()[-1]) == json.loads(out[1].stdout.splitlines()[-1])
        assert json.loads(out[0].stdout.splitlines()[-1])["points"] > 0


def test_hope_node_over_rccl(oracle_lib, tmp_path, monkeypatch):
    """`hope_node <file> ply` with HOPE_NODE_GPUS=1: the C++ multi-GPU host (hope_amd/host/multi_gpu.cpp over
    rccl_comm.cpp: ncclCommInitAll, all-reduces, the all-to-all of routed points, the all-gather of the down-sampled
    cloud -- all real RCCL calls, one rank being what a one-GPU box can hold; 2 and 3 ranks of the same orchestration
    run on threads in tests/test_multi_host_cpu.py).  Records == the oracle's on the whole cloud."""
    from test_host_cpu import check_record, run_node
    from test_sharding_cpu import _layer_cloud
    node = os.path.join(ROOT, "hope_amd", "host", "hope_node")
    xyz = _layer_cloud()
    p = tmp_path / "layers.ply"
    p.write_bytes(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                   "end_header\n" % len(xyz)).encode() + xyz.tobytes())
    monkeypatch.setenv("HOPE_NODE_GPUS", "1")
    r, recs = run_node(node, [p, "ply"], tmp_path, hull=False)
    assert r.returncode == 0, r.stderr + r.stdout
    assert recs[0]["gpus"] == 1 and recs[0]["path"] == "one rank"
    c = HopeContext(oracle_lib)
    c.segment_cloud(xyz, apply_filter_transform=False)
    planes = [k for k in c.clusters(0) if k["pass"] == 1]
    assert len(planes) >= 4
    check_record(recs[0], c.counts(0), planes)


@pytest.mark.parametrize("ranks", [2, 3])
def test_hope_node_ranks_on_one_gpu(oracle_lib, tmp_path, monkeypatch, ranks):
    """The C++ orchestration with MORE than one rank on the card: HOPE_NODE_RANKS_ON_GPU0=N runs N ranks as threads,
    each with its own ctx and device buffers on device 0, collectives as device copies (rccl_comm.cpp
    DeviceThreadComm).  Everything of the multi-rank path but the RCCL transport runs on the HIP library: the staged
    entry points with device pointers, the routed all-to-all, the z bands.  Records == the oracle's on the whole
    cloud."""
    from test_host_cpu import check_record, run_node
    from test_sharding_cpu import _layer_cloud
    node = os.path.join(ROOT, "hope_amd", "host", "hope_node")
    xyz = _layer_cloud()
    p = tmp_path / "layers.ply"
    p.write_bytes(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                   "end_header\n" % len(xyz)).encode() + xyz.tobytes())
    monkeypatch.setenv("HOPE_NODE_RANKS_ON_GPU0", str(ranks))
    r, recs = run_node(node, [p, "ply"], tmp_path, hull=False)
    assert r.returncode == 0, r.stderr + r.stdout
    assert recs[0]["gpus"] == ranks and recs[0]["path"] == "z-bands", recs[0]
    c = HopeContext(oracle_lib)
    c.segment_cloud(xyz, apply_filter_transform=False)
    planes = [k for k in c.clusters(0) if k["pass"] == 1]
    assert len(planes) >= 4
    check_record(recs[0], c.counts(0), planes)
    if ranks == 2:   # a ramp has no empty layer to cut at: the replicated fallback, also on device buffers
        from test_sharding_cpu import _ramp_cloud
        xyz = _ramp_cloud()
        p.write_bytes(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                       "end_header\n" % len(xyz)).encode() + xyz.tobytes())
        r, recs = run_node(node, [p, "ply"], tmp_path, hull=False)
        assert r.returncode == 0, r.stderr + r.stdout
        assert recs[0]["path"] == "replicated", recs[0]
        c = HopeContext(oracle_lib)
        c.segment_cloud(xyz, apply_filter_transform=False)
        check_record(recs[0], c.counts(0), [k for k in c.clusters(0) if k["pass"] == 1])
