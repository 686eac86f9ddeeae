"""Stage-by-stage comparison of two implementations of the C ABI (oracle vs HIP)."""
import numpy as np


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_frame_equal(a, b, frame, check_transformed=False):
    """a, b: HopeContext objects that just ran the same input.  Bit-exact on every stage."""
    ca, cb = a.counts(frame), b.counts(frame)
    for k in ("n_in", "n_raw", "n_vox", "n_fit", "n_segments", "n_clusters", "n_planes", "status", "vox_passthrough"):
        assert getattr(ca, k) == getattr(cb, k), (frame, k, getattr(ca, k), getattr(cb, k))
    if ca.n_raw > 0:
        assert list(ca.grid) == list(cb.grid), (frame, list(ca.grid), list(cb.grid))
        assert list(ca.min_b) == list(cb.min_b)
    if check_transformed:
        xa, ia = a.transformed(frame)
        xb, ib = b.transformed(frame)
        assert np.array_equal(ia, ib)
        assert np.array_equal(bits(xa), bits(xb))          # A1-A3: stricter than the 1e-5 rel. bar
    assert np.array_equal(bits(a.downsampled(frame)), bits(b.downsampled(frame)))   # A5
    na, nb = a.normals(frame), b.normals(frame)
    assert np.array_equal(np.isnan(na), np.isnan(nb))
    assert np.array_equal(bits(na), bits(nb)), np.nanmax(np.abs(na - nb))             # A6
    assert np.array_equal(a.fit_indices(frame), b.fit_indices(frame))                 # A7
    assert np.array_equal(a.knn(frame), b.knn(frame))                                 # A8
    assert np.array_equal(a.labels(frame), b.labels(frame))                           # A9
    cla, clb = a.clusters(frame), b.clusters(frame)
    assert cla.tobytes() == clb.tobytes(), (cla, clb)                                 # A10-A12
    for ci in range(len(cla)):
        assert np.array_equal(a.cluster_indices(frame, ci, int(cla["size"][ci])),
                              b.cluster_indices(frame, ci, int(clb["size"][ci])))
    ma, mb = a.max_plane(frame), b.max_plane(frame)
    assert (ma.valid, ma.n_points, ma.n_members) == (mb.valid, mb.n_points, mb.n_members)
    if ma.valid:
        assert ma.z == mb.z
        assert a.max_plane_members(frame) == b.max_plane_members(frame)
        assert list(ma.members)[:min(ma.n_members, 16)] == a.max_plane_members(frame)[:16]


def summary(ctx, frame):
    """Small JSON-able digest of one frame's results (used for the golden fixtures)."""
    import hashlib
    c = ctx.counts(frame)
    cl = ctx.clusters(frame)
    def h(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
    return {
        "n_raw": int(c.n_raw), "n_vox": int(c.n_vox), "n_fit": int(c.n_fit),
        "n_segments": int(c.n_segments), "n_clusters": int(c.n_clusters), "n_planes": int(c.n_planes),
        "grid": list(c.grid), "min_b": list(c.min_b),
        "sha_downsampled": h(ctx.downsampled(frame)), "sha_normals": h(ctx.normals(frame)),
        "sha_fit_indices": h(ctx.fit_indices(frame)), "sha_knn": h(ctx.knn(frame)),
        "sha_labels": h(ctx.labels(frame)),
        "planes": [[int(r["size"]), float(r["z_mean"]), float(r["min_x"]), float(r["min_y"]),
                    float(r["max_x"]), float(r["max_y"])] for r in cl if r["pass"] == 1],
    }
