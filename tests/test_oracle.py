"""The CPU oracle against (a) the input-side anchors SURVEY.md §8c measured on the reference's
example pair, (b) its own committed golden digests, (c) its brute-force neighbour search and
(d) a build that calls glibc's float trig instead of include/hope_detmath.h.

PARITY UNPINNED: the reference ships no expected outputs for this path, so nothing here compares
with the reference's own results; see DESIGN.md."""
import json
import os

import numpy as np

from conftest import GOLDEN
from parity import assert_frame_equal, summary

from hope_amd.capi import HOPE_MODE_RT, HopeContext


def golden():
    return json.load(open(os.path.join(GOLDEN, "oracle_golden.json")))


def run_example(lib, tum_example, **kw):
    depth, t, q = tum_example
    c = HopeContext(lib, **kw)
    c.set_pose(t, q)
    assert c.segment_depth(depth) == 0
    return c


def test_example_pair_anchors(oracle_lib, tum_example):
    """210 004 valid pixels, bbox, 56x46x79 grid, 5 699 voxels."""
    c = run_example(oracle_lib, tum_example)
    k = c.counts(0)
    assert k.n_in == 307200 and k.n_raw == 210004 and k.n_vox == 5699
    assert list(k.grid) == [56, 46, 79]
    xyz, src = c.transformed(0)
    assert np.allclose(xyz.min(0), [-2.238, 0.443, -0.019], atol=1e-3)
    assert np.allclose(xyz.max(0), [0.511, 2.660, 1.542], atol=1e-3)
    assert np.all(np.diff(src) > 0)
    # ~13 patches in doc/demo.png: floor, desk, monitor tops ...
    cl = c.clusters(0)
    assert k.n_planes == int((cl["pass"] == 1).sum()) == 13
    # the desk: densest z bins of the example are 0.76-0.90 m
    big = cl[np.argmax(cl["size"])]
    assert 0.76 < big["z_mean"] < 0.90 and big["pass"] == 1


def test_transform_matches_float64_rotation(oracle_lib, tum_example):
    """A1-A3 against an independent float64 numpy evaluation: within 1e-5 relative."""
    depth, t, q = tum_example
    c = run_example(oracle_lib, tum_example)
    xyz, src = c.transformed(0)
    z = depth.reshape(-1).astype(np.float64)[src] / 5000.0
    col, row = src % 640, src // 640
    p = np.stack([(col - 318.6) * z / 517.0, (row - 255.3) * z / 517.0, z], 1)
    x, y, zz, w = [float(v) for v in q]
    R = np.array([[1 - 2 * (y * y + zz * zz), 2 * (x * y - zz * w), 2 * (x * zz + y * w)],
                  [2 * (x * y + zz * w), 1 - 2 * (x * x + zz * zz), 2 * (y * zz - x * w)],
                  [2 * (x * zz - y * w), 2 * (y * zz + x * w), 1 - 2 * (x * x + y * y)]])
    ref = p @ R.T + t.astype(np.float64)
    assert np.max(np.abs(xyz - ref) / np.maximum(np.abs(ref), 1.0)) < 1e-5


def rpy_pose(roll, pitch, yaw):
    """tf2::Quaternion::setRPY in double (libm through math), y negated, narrowed to float; translation (0, 0, 1):
    the pose of transform.cpp:108-130, written independently of the libraries under test."""
    import math
    hr, hp, hy = 0.5 * float(np.float32(roll)), 0.5 * float(np.float32(pitch)), 0.5 * float(np.float32(yaw))
    cr, sr, cp, sp, cy, sy = math.cos(hr), math.sin(hr), math.cos(hp), math.sin(hp), math.cos(hy), math.sin(hy)
    q = [sr * cp * cy - cr * sp * sy, -(cr * sp * cy + sr * cp * sy), cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
    return np.float32([0, 0, 1]), np.float32(q)


def rpy_scene(roll, pitch, yaw, n=60000):
    """A sensor-frame cloud that the roll/pitch/yaw pose brings back to a scene of horizontal surfaces."""
    from hope_amd import synth
    world = synth.make_multilevel_cloud(n, first_index=0, device="cpu").numpy().astype(np.float64) * 0.25 + [0, 0, 0.2]
    t, q = rpy_pose(roll, pitch, yaw)
    x, y, z, w = [float(v) for v in q]
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return ((world - t.astype(np.float64)) @ R).astype(np.float32)     # R^T (p - t), row vectors


def test_rpy_pose(oracle_lib):
    """hope_set_pose_rpy (transform.cpp:108-130): known answers of tf2's setRPY, and the same results as
    hope_set_pose with the quaternion computed here."""
    pts = np.float32([[0.5, 0.25, 1.0], [-1.0, 2.0, 3.0], [0.125, -0.5, 0.75]])
    c = HopeContext(oracle_lib)
    c.set_pose_rpy(0.0, 0.0, 0.0)                                       # identity rotation: only the 1 m lift
    c.segment_cloud(pts, apply_filter_transform=True)
    assert np.array_equal(c.transformed(0)[0], pts + np.float32([0, 0, 1]))
    c.set_pose_rpy(0.0, 0.0, np.pi / 2)                                 # yaw 90 deg: (x, y) -> (-y, x)
    c.segment_cloud(pts, apply_filter_transform=True)
    assert np.allclose(c.transformed(0)[0], np.stack([-pts[:, 1], pts[:, 0], pts[:, 2] + 1], 1), atol=1e-6)
    c.set_pose_rpy(0.0, np.pi / 2, 0.0)                                 # pitch 90 deg with y negated: (x, z) -> (-z, x)
    c.segment_cloud(pts, apply_filter_transform=True)
    assert np.allclose(c.transformed(0)[0], np.stack([-pts[:, 2], pts[:, 1], pts[:, 0] + 1], 1), atol=1e-6)
    c.set_pose_rpy(np.pi / 2, 0.0, 0.0)                                 # roll 90 deg: (y, z) -> (-z, y)
    c.segment_cloud(pts, apply_filter_transform=True)
    assert np.allclose(c.transformed(0)[0], np.stack([pts[:, 0], -pts[:, 2], pts[:, 1] + 1], 1), atol=1e-6)
    rpy = (0.3, -0.2, 0.5)
    xyz = rpy_scene(*rpy)
    a, b = HopeContext(oracle_lib), HopeContext(oracle_lib)
    a.set_pose_rpy(*rpy)
    b.set_pose(*rpy_pose(*rpy))
    for k in (a, b):
        k.segment_cloud(xyz, apply_filter_transform=True)
    assert_frame_equal(a, b, 0, check_transformed=True)
    assert a.counts(0).n_planes >= 3                                    # the surfaces are horizontal again


def test_against_committed_golden(oracle_lib, tum_example):
    g = golden()
    assert summary(run_example(oracle_lib, tum_example), 0) == g["tum_example"]
    z = np.load(os.path.join(GOLDEN, "synth_frames_0_2.npz"))
    c = HopeContext(oracle_lib)
    c.set_frame_poses(z["t"], z["q"])
    assert c.segment_depth(z["depth"]) == 0
    for f in range(3):
        assert summary(c, f) == g["synth_frames"][f]
    c2 = HopeContext(oracle_lib)
    c2.segment_cloud(np.load(os.path.join(GOLDEN, "loft_120k.npz"))["xyz"])
    assert summary(c2, 0) == g["loft_120k"]
    c3 = HopeContext(oracle_lib, mode=HOPE_MODE_RT, min_height=0.6, max_height=0.9)
    c3.set_frame_poses(z["t"], z["q"])
    c3.segment_depth(z["depth"])
    m = c3.max_plane(0)
    assert {"valid": m.valid, "n_points": m.n_points, "z": float(m.z), "members": c3.max_plane_members(0),
            "n_planes": c3.counts(0).n_planes} == g["rt_frame0"]


def test_kdtree_equals_bruteforce(oracle_lib, tum_example):
    a = run_example(oracle_lib, tum_example)
    depth, t, q = tum_example
    b = HopeContext(oracle_lib)
    b.lib.dll.hope_oracle_set_bruteforce(b._h, 1)
    b.set_pose(t, q)
    b.segment_depth(depth)
    assert_frame_equal(a, b, 0)


def test_glibc_trig_gives_the_same_planes(oracle_lib, oracle_libm_lib, tum_example):
    """Sensitivity of the result to the last bit of atan2f/cosf/sinf/asinf/acosf: none on the
    example pair (labels and plane records identical; normals within 1e-6)."""
    a = run_example(oracle_lib, tum_example)
    b = run_example(oracle_libm_lib, tum_example)
    assert np.array_equal(a.fit_indices(0), b.fit_indices(0))
    assert np.array_equal(a.labels(0), b.labels(0))
    ca, cb = a.clusters(0), b.clusters(0)
    assert np.array_equal(ca["size"], cb["size"]) and np.array_equal(ca["pass"], cb["pass"])
    assert np.array_equal(ca["z_mean"], cb["z_mean"])
    na, nb = a.normals(0), b.normals(0)
    ok = np.isfinite(na).all(1)
    # near-degenerate neighbourhoods amplify the last bit of the root; typical difference ~1e-7
    assert np.median(np.abs(na[ok] - nb[ok])) < 1e-6


def test_edge_cases(oracle_lib):
    c = HopeContext(oracle_lib)
    assert c.segment_depth(np.zeros((2, 480, 640), np.uint16)) == 1          # HOPE_EMPTY
    assert c.counts(1).n_raw == 0 and c.counts(1).n_vox == 0 and c.counts(1).status == 1
    # out-of-range depths only (>= 8 m, <= 0.3 m)
    d = np.full((1, 480, 640), 40000, np.uint16)
    d[0, :10] = 1500
    assert c.segment_depth(d) == 1
    # two points: no normal has >= 3 neighbours -> nothing passes the gate
    assert c.segment_cloud(np.array([[0, 0, 0], [0.01, 0, 0]], np.float32)) == 2
    k = c.counts(0)
    assert (k.n_raw, k.n_vox, k.n_fit, k.n_clusters) == (2, 1, 0, 0)
    # a tiny flat patch: fewer fit points than the 8 neighbours asked for
    g = np.stack(np.meshgrid(np.arange(2) * 0.05 + 0.02, np.arange(3) * 0.05 + 0.02), -1).reshape(-1, 2)
    pts = np.concatenate([g, np.full((6, 1), 0.5)], 1).astype(np.float32)
    st = c.segment_cloud(pts)
    k = c.counts(0)
    assert st == 0 and k.n_vox == 6 and k.n_fit == 6
    nb = c.knn(0)
    assert (nb[:, :6] >= 0).all() and (nb[:, 6:] == -1).all()
    assert k.n_clusters == 1 and c.clusters(0)["size"][0] == 6
    # NaN / inf records are dropped; PassThrough limits are inclusive
    cfgc = HopeContext(oracle_lib)
    pts = np.array([[0, 0, 0.3], [0, 0, 8.0], [0, 0, 8.0001], [np.nan, 0, 1], [0, np.inf, 1]], np.float32)
    cfgc.segment_cloud(pts, apply_filter_transform=True)
    assert cfgc.counts(0).n_raw == 2


def test_rt_height_window_and_merge(oracle_lib):
    z = np.load(os.path.join(GOLDEN, "synth_frames_0_2.npz"))
    c = HopeContext(oracle_lib, mode=HOPE_MODE_RT, min_height=0.6, max_height=0.9)
    c.set_frame_poses(z["t"], z["q"])
    c.segment_depth(z["depth"])
    cl = c.clusters(0)
    m = c.max_plane(0)
    assert m.valid == 1 and 0.6 < m.z < 0.9
    inside = [i for i in range(len(cl)) if cl["pass"][i] and 0.6 < cl["z_mean"][i] < 0.9]
    assert c.counts(0).n_planes == len(inside)
    assert set(c.max_plane_members(0)) <= set(inside)
    assert m.n_points == sum(int(cl["size"][i]) for i in c.max_plane_members(0))
    # dynamic reconfigure analogue: move the window to the floor
    c.set_height_window(-0.1, 0.3)
    c.segment_depth(z["depth"])
    m2 = c.max_plane(0)
    assert m2.valid == 1 and abs(m2.z) < 0.05 and m2.n_points > m.n_points
    c.set_height_window(3.0, 4.0)
    c.segment_depth(z["depth"])
    assert c.max_plane(0).valid == 0


def test_rt_winner_merges_more_than_sixteen_clusters(oracle_lib):
    """plane_segment.cpp:943-960: every passing cluster within th_z of the current max plane is merged into
    it, however many there are; the winner's member list and point count are those of all of them."""
    from conftest import patch_cloud
    xyz = patch_cloud(24)
    c = HopeContext(oracle_lib, mode=HOPE_MODE_RT, min_height=0.8, max_height=1.5, aggressive_merge=1)
    c.segment_cloud(xyz, apply_filter_transform=True)
    cl = c.clusters(0)
    m = c.max_plane(0)
    mem = c.max_plane_members(0)
    passing = [i for i in range(len(cl)) if cl["pass"][i] and 0.8 < cl["z_mean"][i] < 1.5]
    assert len(passing) >= 20 and m.valid == 1
    assert m.n_members == len(mem) == len(passing)               # all at one height: all merged
    assert mem == passing[::-1]                                  # combineCloud: newest cluster first
    assert m.n_points == sum(int(cl["size"][i]) for i in mem)
    assert list(m.members) == mem[:16]
    # without the merge the winner is the largest single cluster
    c2 = HopeContext(oracle_lib, mode=HOPE_MODE_RT, min_height=0.8, max_height=1.5, aggressive_merge=0)
    c2.segment_cloud(xyz, apply_filter_transform=True)
    assert c2.max_plane(0).n_members == 1 and c2.max_plane(0).n_points == max(int(cl["size"][i]) for i in passing)


def test_voxelgrid_leaf_too_small_passes_the_input_through(oracle_lib):
    """PCL 1.8.1 voxel_grid.hpp applyFilter: (dx*dy*dz) > INT_MAX -> warning, output = *input.  Two level
    patches 300 m apart at the default leaf (6000 x 6000 x 400 cells): the down-sampled cloud is the
    input, point for point, and both patches still come out as planes."""
    rng = np.random.default_rng(5)
    a = np.stack([rng.uniform(0, 0.4, 600), rng.uniform(0, 0.4, 600), rng.normal(0.7, 0.002, 600)], 1)
    b = a + [300.0, 300.0, 8.0]
    xyz = np.concatenate([a, b]).astype(np.float32)
    c = HopeContext(oracle_lib)
    assert c.segment_cloud(xyz) == 0
    k = c.counts(0)
    assert k.vox_passthrough == 1 and k.n_vox == k.n_raw == 1200 and list(k.grid) == [0, 0, 0]
    assert np.array_equal(c.downsampled(0).view(np.uint32), xyz.view(np.uint32))
    assert k.n_planes == 2
    # the same patches side by side fit the grid and are down-sampled
    near = np.concatenate([a, a + [1.0, 0.0, 0.3]]).astype(np.float32)
    assert c.segment_cloud(near) == 0
    assert c.counts(0).vox_passthrough == 0 and c.counts(0).n_vox < 300


def test_fixed_orders_move_plane_heights_by_less_than_a_millimetre():
    """tools/order_study.py: flipping any one of the orders the oracle fixes (voxel sums, neighbour ties,
    covariance sums) to another legal order keeps the cluster count, and moves no plane height by 1 mm."""
    import subprocess
    import sys
    from conftest import ROOT
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "variants"])
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import order_study
    z = np.load(os.path.join(GOLDEN, "synth_frames_0_2.npz"))
    g = np.load(os.path.join(GOLDEN, "tum_example_depth.npz"))
    inputs = [(g["depth"], g["t"], g["q"])] + [(z["depth"][i], z["t"][i], z["q"][i]) for i in range(3)]
    from hope_amd.capi import HopeLib
    base = order_study.run(HopeLib(os.path.join(order_study.BUILD, "libhope_oracle.so"), "hope_oracle_"), inputs)
    for tag, _ in order_study.VARIANTS:
        var = order_study.run(HopeLib(os.path.join(order_study.BUILD, f"libhope_oracle_{tag}.so"), "hope_oracle_"), inputs)
        r = order_study.compare(base, var)
        assert r["cl_diff"] == 0 and r["unmatched"] == 0 and r["dz_mm"] < 1.0 and r["max_m"] < 1e-5, (tag, r)
        assert abs(r["planes_b"] - r["planes_v"]) <= 1, (tag, r)
