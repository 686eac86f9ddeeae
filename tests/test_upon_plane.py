"""Utilities::getClustersUponPlane (filter above the winning plane inside its
contour + pcl::EuclideanClusterExtraction).  The oracle against an independent scipy evaluation on the CPU;
the HIP path against the oracle on the GPU (bit-exact: same points, same clusters, same order)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, synth_frames
from test_host_cpu import host, hull  # noqa: F401  (fixture + helper)

from hope_amd.capi import HOPE_MODE_RT, HopeContext


def winner_contour(ctx, host_lib, frame=0):
    """max_plane_contour_: hull of the RT winner's points (plane_segment.cpp:943-957)."""
    m = ctx.max_plane(frame)
    assert m.valid
    vox, fit, cl = ctx.downsampled(frame), ctx.fit_indices(frame), ctx.clusters(frame)
    pts = np.concatenate([vox[fit[ctx.cluster_indices(frame, ci, int(cl["size"][ci]))]] for ci in ctx.max_plane_members(frame)])
    idx, _ = hull(host_lib, pts)
    return pts[idx]


def rt_ctx(lib, lo=0.5, hi=1.5):
    return HopeContext(lib, mode=HOPE_MODE_RT, min_height=lo, max_height=hi)


def components_f32(xyz, tol):
    """Connected components of dist2 < float(tol*tol) with the distance in float32, FLANN's operation order."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    from scipy.spatial import cKDTree
    r2 = np.float32(np.float64(np.float32(tol)) ** 2)
    pairs = cKDTree(xyz.astype(np.float64)).query_pairs(float(tol) * 1.01, output_type="ndarray")
    d = xyz[pairs[:, 0]] - xyz[pairs[:, 1]]
    d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    pairs = pairs[d2 < r2]
    g = coo_matrix((np.ones(len(pairs)), (pairs[:, 0], pairs[:, 1])), shape=(len(xyz), len(xyz)))
    return connected_components(g, directed=False)[1]


def test_oracle_upon_plane_against_scipy(oracle_lib, host, tum_example):
    depth, t, q = tum_example
    c = rt_ctx(oracle_lib)
    c.set_pose(t, q)
    c.segment_depth(depth)
    contour = winner_contour(c, host)
    n_upon, all_pts = c.clusters_upon_plane(contour, min_size=1)
    assert n_upon > 1000
    xyz = np.concatenate([p for p, _ in all_pts])
    src = np.concatenate([s for _, s in all_pts])
    assert len(xyz) == n_upon and len(set(src)) == n_upon
    # the filter: every kept point is above the contour's top and inside its bounding rectangle and polygon;
    # every raw point clearly inside (1 cm from the border) and above is kept
    raw, raw_src = c.transformed(0)
    z_thr = np.float64(contour[:, 2].max()) + 0.01
    assert np.all(xyz[:, 2].astype(np.float64) >= z_thr)
    from matplotlib.path import Path
    poly = Path(contour[:, :2].astype(np.float64))
    sure_in = poly.contains_points(raw[:, :2].astype(np.float64), radius=-0.02) & poly.contains_points(raw[:, :2].astype(np.float64), radius=0.02)
    sure_in &= raw[:, 2].astype(np.float64) >= z_thr
    assert set(raw_src[sure_in]) <= set(src)
    far_out = ~(poly.contains_points(raw[:, :2].astype(np.float64), radius=-0.02) | poly.contains_points(raw[:, :2].astype(np.float64), radius=0.02))
    assert not (set(raw_src[far_out]) & set(src))
    # the clustering: partition == connected components of the float32 radius graph over `temp`
    order = np.argsort(src)
    temp = xyz[order]
    lab = components_f32(temp, 0.01)
    pos = {s: i for i, s in enumerate(src[order])}
    for p, s in all_pts:
        ids = [pos[v] for v in s]
        assert ids == sorted(ids)                                  # indices ascending inside a cluster
        assert len(set(lab[ids])) == 1 and (lab == lab[ids[0]]).sum() == len(ids)
    sizes = [len(s) for _, s in all_pts]
    assert sizes == sorted(sizes, reverse=True) and len(sizes) == lab.max() + 1
    # the reference's own gates: 10 <= size <= 240000
    n2, kept = c.clusters_upon_plane(contour)
    assert n2 == n_upon and [len(s) for _, s in kept] == [v for v in sizes if 10 <= v <= 240000]
    assert len(kept) >= 1                                          # there are objects on that desk
    # an empty contour keeps nothing; a contour above everything keeps nothing
    assert c.clusters_upon_plane(np.zeros((0, 3), np.float32)) == (0, [])
    assert c.clusters_upon_plane(contour + np.float32([0, 0, 50]))[0] == 0


def run_both(oracle_lib, hip_lib, host_lib, segment, frames=(0,), lo=0.5, hi=1.5, variants=((0.01, 10, 240000), (0.01, 1, 240000), (0.03, 3, 50))):
    a, b = rt_ctx(oracle_lib, lo, hi), rt_ctx(hip_lib, lo, hi)
    for c in (a, b):
        segment(c)
    total = 0
    for f in frames:
        if not a.max_plane(f).valid:
            continue
        contour = winner_contour(a, host_lib, f)
        assert np.array_equal(contour, winner_contour(b, host_lib, f))
        for tol, mn, mx in variants:
            na, ca = a.clusters_upon_plane(contour, frame=f, tolerance=tol, min_size=mn, max_size=mx)
            nb, cb = b.clusters_upon_plane(contour, frame=f, tolerance=tol, min_size=mn, max_size=mx)
            assert na == nb and len(ca) == len(cb), (f, tol, mn, mx, na, nb, len(ca), len(cb))
            for (pa, sa), (pb, sb) in zip(ca, cb):
                assert np.array_equal(sa, sb) and pa.tobytes() == pb.tobytes()
            total += len(ca)
    return total


@pytest.mark.gpu
def test_upon_plane_example_pair(oracle_lib, hip_lib, host, tum_example):
    depth, t, q = tum_example

    def seg(c):
        c.set_pose(t, q)
        assert c.segment_depth(depth) == 0
    assert run_both(oracle_lib, hip_lib, host, seg) >= 3


@pytest.mark.gpu
def test_upon_plane_synthetic_frames_and_cloud_input(oracle_lib, hip_lib, host):
    d, t, q = synth_frames(6)

    def seg(c):
        c.set_frame_poses(t, q)
        assert c.segment_depth(d) == 0
    assert run_both(oracle_lib, hip_lib, host, seg, frames=range(6)) >= 6        # the boxes on the table tops
    # the same through the cloud entry (PlaneSegmentRT::cloudCallback path), sensor-frame cloud + pose
    z = d[1].astype(np.float32) / np.float32(5000.0)
    cols = (np.arange(640, dtype=np.float32) - np.float32(318.6))[None, :]
    rows = (np.arange(480, dtype=np.float32) - np.float32(255.3))[:, None]
    cloud = np.stack([cols * z / np.float32(517.0), rows * z / np.float32(517.0), z], -1).reshape(-1, 3)
    cloud[z.reshape(-1) == 0] = np.nan

    def seg_cloud(c):
        c.set_pose(t[1], q[1])
        assert c.segment_cloud(cloud, apply_filter_transform=True) == 0
    assert run_both(oracle_lib, hip_lib, host, seg_cloud) >= 1
    # floor as the winner: everything standing on it
    assert run_both(oracle_lib, hip_lib, host, seg, frames=range(2), lo=-0.2, hi=0.3, variants=((0.01, 10, 240000),)) >= 1
