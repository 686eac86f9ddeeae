"""ctypes binding of the C ABI declared in include/hope_gpu.h.

The same binding drives the HIP library (prefix ``hope_``, hope_amd/csrc/libhope_hip.so)
and, in tests and the bench's cpu_baseline leg only, the CPU oracle (prefix
``hope_oracle_``, oracle/_build/libhope_oracle.so).  Nothing in this package loads the
oracle: callers that want it pass its path and prefix explicitly.
"""
import ctypes as C
import os

import numpy as np

HOPE_OK = 0
HOPE_EMPTY = 1
HOPE_NO_NORMAL_FIT = 2
HOPE_MODE_OFFLINE = 0
HOPE_MODE_RT = 1

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP_LIB_PATH = os.path.join(REPO_ROOT, "hope_amd", "csrc", "libhope_hip.so")


class HopeConfig(C.Structure):
    _fields_ = [
        ("th_xy", C.c_float), ("th_z", C.c_float),
        ("min_depth", C.c_float), ("max_depth", C.c_float),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("depth_scale", C.c_float),
        ("knn", C.c_int32), ("min_cluster", C.c_int32), ("mode", C.c_int32),
        ("min_height", C.c_float), ("max_height", C.c_float),
        ("aggressive_merge", C.c_int32), ("device", C.c_int32), ("sub_batches", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


class HopeCounts(C.Structure):
    _fields_ = [
        ("n_in", C.c_int64), ("n_raw", C.c_int64), ("n_vox", C.c_int64), ("n_fit", C.c_int64),
        ("n_segments", C.c_int32), ("n_clusters", C.c_int32), ("n_planes", C.c_int32),
        ("grid", C.c_int32 * 3), ("min_b", C.c_int32 * 3),
        ("vox_passthrough", C.c_int32), ("status", C.c_int32),
        ("reserved", C.c_int32 * 3),
    ]


class HopeCluster(C.Structure):
    _fields_ = [
        ("size", C.c_int32), ("pass_", C.c_int32), ("z_mean", C.c_float),
        ("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
        ("seed_rank", C.c_int32),
    ]


class HopeMaxPlane(C.Structure):
    _fields_ = [
        ("valid", C.c_int32), ("n_points", C.c_int32), ("z", C.c_float),
        ("n_members", C.c_int32), ("members", C.c_int32 * 16),
    ]


CLUSTER_DTYPE = np.dtype([
    ("size", "<i4"), ("pass", "<i4"), ("z_mean", "<f4"),
    ("min_x", "<f4"), ("min_y", "<f4"), ("max_x", "<f4"), ("max_y", "<f4"),
    ("seed_rank", "<i4"),
])
assert CLUSTER_DTYPE.itemsize == C.sizeof(HopeCluster)

# every symbol include/hope_gpu.h declares (without prefix)
ABI_SYMBOLS = [
    "config_default", "create", "destroy", "last_error", "set_pose", "set_pose_rpy", "set_frame_poses",
    "set_height_window", "segment_depth_u16", "segment_cloud_xyz", "num_frames",
    "frame_counts", "get_transformed", "get_downsampled", "get_normals", "get_fit_indices",
    "get_knn", "get_labels", "get_clusters", "get_cluster_indices", "get_cluster_hull_candidates", "get_max_plane", "get_max_plane_members",
    "last_timing", "grid_from_bbox", "shard_bbox", "shard_z_histogram", "shard_route",
    "downsample_cloud_bbox", "segment_downsampled", "segment_downsampled_staged",
    "get_normals_rows", "get_knn_rows", "get_downsampled_to", "count_band_crossings",
    "clusters_upon_plane", "get_upon_cluster_sizes", "get_upon_cluster",
]


class HopeError(RuntimeError):
    pass


class HopeLib:
    """A loaded implementation of the C ABI."""

    def __init__(self, path=HIP_LIB_PATH, prefix="hope_"):
        if not os.path.exists(path):
            raise HopeError(
                f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C hope_amd/csrc). There is no CPU fallback.")
        self.path = path
        self.prefix = prefix
        if prefix == "hope_":
            # A PyTorch-ROCm wheel brings a HIP runtime of its own, and a process has to stay with ONE runtime: when the
            # library is loaded first it pulls in /opt/rocm's, and a later `import torch` then finds "No HIP GPUs".
            # Torch first, and the library binds to the runtime that is already there.  (C/C++ hosts link /opt/rocm.)
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        self.dll = C.CDLL(path, mode=C.RTLD_LOCAL)
        vp, i32, f32p = C.c_void_p, C.c_int, C.POINTER(C.c_float)
        sig = {
            "config_default": (C.c_int, [C.POINTER(HopeConfig)]),
            "create": (C.c_int, [C.POINTER(HopeConfig), C.POINTER(vp)]),
            "destroy": (None, [vp]),
            "last_error": (C.c_char_p, [vp]),
            "set_pose": (C.c_int, [vp, f32p, f32p]),
            "set_pose_rpy": (C.c_int, [vp, C.c_float, C.c_float, C.c_float]),
            "set_frame_poses": (C.c_int, [vp, i32, f32p, f32p]),
            "set_height_window": (C.c_int, [vp, C.c_float, C.c_float]),
            "segment_depth_u16": (C.c_int, [vp, vp, i32, i32, i32, i32]),
            "segment_cloud_xyz": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, i32, i32]),
            "num_frames": (C.c_int, [vp]),
            "frame_counts": (C.c_int, [vp, i32, C.POINTER(HopeCounts)]),
            "get_transformed": (C.c_int, [vp, i32, vp, vp]),
            "get_downsampled": (C.c_int, [vp, i32, vp]),
            "get_normals": (C.c_int, [vp, i32, vp]),
            "get_fit_indices": (C.c_int, [vp, i32, vp]),
            "get_knn": (C.c_int, [vp, i32, vp]),
            "get_labels": (C.c_int, [vp, i32, vp]),
            "get_clusters": (C.c_int, [vp, i32, vp, i32]),
            "get_cluster_indices": (C.c_int, [vp, i32, i32, vp, i32]),
            "get_cluster_hull_candidates": (C.c_int, [vp, i32, i32, vp, i32]),
            "get_max_plane": (C.c_int, [vp, i32, C.POINTER(HopeMaxPlane)]),
            "get_max_plane_members": (C.c_int, [vp, i32, vp, i32]),
            "last_timing": (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
            "grid_from_bbox": (C.c_int, [vp, vp, vp, vp]),
            "shard_bbox": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, i32, i32, vp, C.POINTER(C.c_int64)]),
            "shard_z_histogram": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, i32, i32, vp, vp, i32]),
            "shard_route": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, i32, i32, vp, vp, i32, i32, vp, i32, vp]),
            "downsample_cloud_bbox": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, i32, vp]),
            "segment_downsampled": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, i32, vp, C.c_int64]),
            "get_downsampled_to": (C.c_int, [vp, i32, vp, i32]),
            "get_normals_rows": (C.c_int, [vp, i32, C.c_int64, C.c_int64, vp, i32]),
            "get_knn_rows": (C.c_int, [vp, i32, C.c_int64, C.c_int64, vp, i32]),
            "count_band_crossings": (C.c_int, [vp, i32, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]),
            "segment_downsampled_staged": (C.c_int, [vp, vp, C.c_size_t, C.c_size_t, i32, vp, C.c_int64, C.c_int64,
                                                     C.c_int64, vp, vp, i32]),
            "clusters_upon_plane": (C.c_int, [vp, i32, vp, i32, C.c_float, i32, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
            "get_upon_cluster_sizes": (C.c_int, [vp, vp, i32]),
            "get_upon_cluster": (C.c_int, [vp, i32, vp, vp, i32]),
        }
        self.fn = {}
        for name, (res, args) in sig.items():
            f = getattr(self.dll, prefix + name)
            f.restype = res
            f.argtypes = args
            self.fn[name] = f
        # instrumentation entry points exist in the HIP library only
        # the no-allocation entry point exists in the HIP library only
        if hasattr(self.dll, prefix + "shard_z_histogram_static"):
            f = getattr(self.dll, prefix + "shard_z_histogram_static")
            f.restype = C.c_int
            f.argtypes = [vp, vp, C.c_size_t, C.c_size_t, i32, vp, vp, i32, i32]
            self.fn["shard_z_histogram_static"] = f
        self.has_profile = hasattr(self.dll, prefix + "profile_enable")
        if self.has_profile:
            f = getattr(self.dll, prefix + "profile_enable")
            f.restype, f.argtypes = C.c_int, [vp, i32]
            self.fn["profile_enable"] = f
            f = getattr(self.dll, prefix + "profile_stage_count")
            f.restype, f.argtypes = C.c_int, []
            self.fn["profile_stage_count"] = f
            f = getattr(self.dll, prefix + "profile_stage")
            f.restype = C.c_int
            f.argtypes = [vp, i32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
            self.fn["profile_stage"] = f
            f = getattr(self.dll, prefix + "profile_stage_bytes")
            f.restype = C.c_int
            f.argtypes = [vp, i32, C.POINTER(C.c_double)]
            self.fn["profile_stage_bytes"] = f
            f = getattr(self.dll, prefix + "fast_division_active")
            f.restype, f.argtypes = C.c_int, [vp]
            self.fn["fast_division_active"] = f
            f = getattr(self.dll, prefix + "debug_chain_sums")
            f.restype, f.argtypes = C.c_int, [vp, vp, C.c_size_t, i32, vp, vp]
            self.fn["debug_chain_sums"] = f
            f = getattr(self.dll, prefix + "debug_xs_stats")
            f.restype, f.argtypes = C.c_int, [vp, vp]
            self.fn["debug_xs_stats"] = f

    def default_config(self):
        cfg = HopeConfig()
        self.fn["config_default"](C.byref(cfg))
        return cfg


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class HopeContext:
    """One hope_ctx.  Mirrors the life cycle of a PlaneSegment / PlaneSegmentRT object."""

    def __init__(self, lib, **overrides):
        self.lib = lib
        cfg = lib.default_config()
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise AttributeError(k)
            setattr(cfg, k, v)
        self.cfg = cfg
        self._h = C.c_void_p()
        st = lib.fn["create"](C.byref(cfg), C.byref(self._h))
        if st != 0:
            raise HopeError(f"hope_create failed: {st}")

    def close(self):
        if self._h:
            self.lib.fn["destroy"](self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st, what):
        if st < 0:
            msg = self.lib.fn["last_error"](self._h)
            raise HopeError(f"{what} failed ({st}): {msg.decode() if msg else ''}")
        return st

    # -- pose / parameters ---------------------------------------------------------------
    def set_pose(self, t, q):
        t, q = _f32(t), _f32(q)
        assert t.shape == (3,) and q.shape == (4,)
        fp = C.POINTER(C.c_float)
        self._check(self.lib.fn["set_pose"](self._h, t.ctypes.data_as(fp), q.ctypes.data_as(fp)), "set_pose")

    def set_pose_rpy(self, roll, pitch, yaw):
        """PlaneSegment::setRPY + the roll/pitch/yaw doTransform (transform.cpp:108-130); radians."""
        self._check(self.lib.fn["set_pose_rpy"](self._h, float(roll), float(pitch), float(yaw)), "set_pose_rpy")

    def set_frame_poses(self, t, q):
        t, q = _f32(t), _f32(q)
        n = t.shape[0]
        assert t.shape == (n, 3) and q.shape == (n, 4)
        fp = C.POINTER(C.c_float)
        self._check(self.lib.fn["set_frame_poses"](self._h, n, t.ctypes.data_as(fp), q.ctypes.data_as(fp)),
                    "set_frame_poses")

    def set_height_window(self, lo, hi):
        self._check(self.lib.fn["set_height_window"](self._h, lo, hi), "set_height_window")

    # -- hot path ------------------------------------------------------------------------
    def segment_depth(self, depth):
        """depth: uint16 numpy array [n_frames, h, w] or [h, w]."""
        d = np.ascontiguousarray(depth, dtype=np.uint16)
        if d.ndim == 2:
            d = d[None]
        n, h, w = d.shape
        return self._check(self.lib.fn["segment_depth_u16"](self._h, d.ctypes.data, w, h, n, 0), "segment_depth_u16")

    def segment_depth_device(self, dev_ptr, w, h, n_frames):
        """depth already resident in HBM (raw device pointer, e.g. torch tensor.data_ptr())."""
        return self._check(self.lib.fn["segment_depth_u16"](self._h, dev_ptr, w, h, n_frames, 1),
                           "segment_depth_u16")

    def segment_cloud(self, xyz, apply_filter_transform=False):
        """xyz: float32 [n, c] with c >= 3 (row stride c*4 bytes)."""
        a = np.ascontiguousarray(xyz, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] >= 3
        return self._check(
            self.lib.fn["segment_cloud_xyz"](self._h, a.ctypes.data, a.shape[0], a.shape[1] * 4,
                                             1 if apply_filter_transform else 0, 0), "segment_cloud_xyz")

    def segment_cloud_device(self, dev_ptr, n, stride_bytes, apply_filter_transform=False):
        return self._check(
            self.lib.fn["segment_cloud_xyz"](self._h, dev_ptr, n, stride_bytes,
                                             1 if apply_filter_transform else 0, 1), "segment_cloud_xyz")

    # -- staged entry points: one cloud across several ranks (hope_amd/sharding.py) ----------
    # ptr: address of n records of stride_bytes bytes, on the host or (on_device) in HBM
    def grid_from_bbox(self, bbox):
        b = np.ascontiguousarray(bbox, dtype=np.float32)
        mn, dv = np.zeros(3, np.int32), np.zeros(3, np.int32)
        st = self._check(self.lib.fn["grid_from_bbox"](self._h, b.ctypes.data, mn.ctypes.data, dv.ctypes.data),
                         "grid_from_bbox")
        return mn, dv, st

    def shard_bbox(self, ptr, n, stride_bytes, apply_filter_transform, on_device):
        bbox = np.zeros(6, np.float32)
        n_raw = C.c_int64(0)
        self._check(self.lib.fn["shard_bbox"](self._h, ptr, n, stride_bytes, int(apply_filter_transform),
                                              int(on_device), bbox.ctypes.data, C.byref(n_raw)), "shard_bbox")
        return bbox, n_raw.value

    def shard_z_histogram(self, ptr, n, stride_bytes, apply_filter_transform, on_device, bbox, n_bins):
        b = np.ascontiguousarray(bbox, dtype=np.float32)
        hist = np.zeros(n_bins, np.int64)
        self._check(self.lib.fn["shard_z_histogram"](self._h, ptr, n, stride_bytes, int(apply_filter_transform),
                                                     int(on_device), b.ctypes.data, hist.ctypes.data, n_bins),
                    "shard_z_histogram")
        return hist

    def shard_z_histogram_static(self, dev_ptr, n, stride_bytes, apply_filter_transform, bbox, n_bins, out_ptr=None):
        """hope_shard_z_histogram_static: device input, no allocation; the bins go to out_ptr (device) or come
        back as a numpy array."""
        b = np.ascontiguousarray(bbox, dtype=np.float32)
        if out_ptr is not None:
            self._check(self.lib.fn["shard_z_histogram_static"](self._h, dev_ptr, n, stride_bytes,
                                                                int(apply_filter_transform), b.ctypes.data, out_ptr,
                                                                n_bins, 1), "shard_z_histogram_static")
            return None
        hist = np.zeros(n_bins, np.int64)
        self._check(self.lib.fn["shard_z_histogram_static"](self._h, dev_ptr, n, stride_bytes,
                                                            int(apply_filter_transform), b.ctypes.data,
                                                            hist.ctypes.data, n_bins, 0), "shard_z_histogram_static")
        return hist

    def shard_route(self, ptr, n, stride_bytes, apply_filter_transform, on_device, bbox, layer_owner, n_ranks,
                    out_ptr, out_on_device):
        b = np.ascontiguousarray(bbox, dtype=np.float32)
        own = np.ascontiguousarray(layer_owner, dtype=np.int32)
        counts = np.zeros(n_ranks, np.int64)
        self._check(self.lib.fn["shard_route"](self._h, ptr, n, stride_bytes, int(apply_filter_transform),
                                               int(on_device), b.ctypes.data, own.ctypes.data, own.shape[0], n_ranks,
                                               out_ptr, int(out_on_device), counts.ctypes.data), "shard_route")
        return counts

    def downsample_cloud_bbox(self, ptr, n, stride_bytes, on_device, bbox):
        b = np.ascontiguousarray(bbox, dtype=np.float32)
        return self._check(self.lib.fn["downsample_cloud_bbox"](self._h, ptr, n, stride_bytes, int(on_device),
                                                                b.ctypes.data), "downsample_cloud_bbox")

    def segment_downsampled(self, ptr, n_vox, stride_bytes, on_device, bbox, n_raw_total):
        b = np.ascontiguousarray(bbox, dtype=np.float32)
        return self._check(self.lib.fn["segment_downsampled"](self._h, ptr, n_vox, stride_bytes, int(on_device),
                                                              b.ctypes.data, int(n_raw_total)), "segment_downsampled")

    # -- results -------------------------------------------------------------------------
    def num_frames(self):
        return self.lib.fn["num_frames"](self._h)

    def counts(self, frame=0):
        c = HopeCounts()
        self._check(self.lib.fn["frame_counts"](self._h, frame, C.byref(c)), "frame_counts")
        return c

    def transformed(self, frame=0):
        n = self.counts(frame).n_raw
        xyz = np.empty((n, 3), np.float32)
        src = np.empty((n,), np.int32)
        self._check(self.lib.fn["get_transformed"](self._h, frame, xyz.ctypes.data, src.ctypes.data), "get_transformed")
        return xyz, src

    def downsampled(self, frame=0):
        n = self.counts(frame).n_vox
        xyz = np.empty((n, 3), np.float32)
        if n:
            self._check(self.lib.fn["get_downsampled"](self._h, frame, xyz.ctypes.data), "get_downsampled")
        return xyz

    def segment_downsampled_staged(self, ptr, n_vox, stride_bytes, on_device, bbox, n_raw_total, own_first,
                                   own_count, normals_ptr=None, knn_ptr=None, stop_after=0):
        """hope_segment_downsampled_staged: A6 / A8 for the own range only, gathered stages passed back in."""
        b = np.ascontiguousarray(bbox, np.float32)
        return self._check(self.lib.fn["segment_downsampled_staged"](
            self._h, ptr, n_vox, stride_bytes, int(on_device), b.ctypes.data, int(n_raw_total), int(own_first),
            int(own_count), normals_ptr, knn_ptr, int(stop_after)), "segment_downsampled_staged")

    def downsampled_to(self, out_ptr, out_on_device, frame=0):
        """The down-sampled cloud (n_vox x 3 floats) to out_ptr, host or device."""
        self._check(self.lib.fn["get_downsampled_to"](self._h, frame, out_ptr, int(out_on_device)), "get_downsampled_to")

    def normals_rows(self, first, count, frame=0, out_ptr=None, out_on_device=False):
        """Rows [first, first+count) of the normals: into a new numpy array, or to out_ptr (host or device)."""
        if out_ptr is not None:
            self._check(self.lib.fn["get_normals_rows"](self._h, frame, int(first), int(count), out_ptr,
                                                        int(out_on_device)), "get_normals_rows")
            return None
        out = np.empty((count, 4), np.float32)
        self._check(self.lib.fn["get_normals_rows"](self._h, frame, int(first), int(count), out.ctypes.data, 0),
                    "get_normals_rows")
        return out

    def knn_rows(self, first, count, frame=0, out_ptr=None, out_on_device=False):
        if out_ptr is not None:
            self._check(self.lib.fn["get_knn_rows"](self._h, frame, int(first), int(count), out_ptr,
                                                    int(out_on_device)), "get_knn_rows")
            return None
        out = np.empty((count, self.cfg.knn), np.int32)
        self._check(self.lib.fn["get_knn_rows"](self._h, frame, int(first), int(count), out.ctypes.data, 0),
                    "get_knn_rows")
        return out

    def count_band_crossings(self, fit_first, fit_count, frame=0):
        """Valid edges (|dz| < th_z) from the fit points [fit_first, fit_first + fit_count) to fit points outside."""
        n = C.c_int64(0)
        self._check(self.lib.fn["count_band_crossings"](self._h, frame, fit_first, fit_count, C.byref(n)),
                    "count_band_crossings")
        return int(n.value)

    def normals(self, frame=0):
        n = self.counts(frame).n_vox
        out = np.empty((n, 4), np.float32)
        if n:
            self._check(self.lib.fn["get_normals"](self._h, frame, out.ctypes.data), "get_normals")
        return out

    def fit_indices(self, frame=0):
        n = self.counts(frame).n_fit
        out = np.empty((n,), np.int32)
        if n:
            self._check(self.lib.fn["get_fit_indices"](self._h, frame, out.ctypes.data), "get_fit_indices")
        return out

    def knn(self, frame=0):
        n = self.counts(frame).n_fit
        out = np.empty((n, self.cfg.knn), np.int32)
        if n:
            self._check(self.lib.fn["get_knn"](self._h, frame, out.ctypes.data), "get_knn")
        return out

    def labels(self, frame=0):
        n = self.counts(frame).n_fit
        out = np.empty((n,), np.int32)
        if n:
            self._check(self.lib.fn["get_labels"](self._h, frame, out.ctypes.data), "get_labels")
        return out

    def clusters(self, frame=0):
        n = self.counts(frame).n_clusters
        out = np.zeros((n,), CLUSTER_DTYPE)
        if n:
            got = self._check(self.lib.fn["get_clusters"](self._h, frame, out.ctypes.data, n), "get_clusters")
            assert got == n
        return out

    def cluster_indices(self, frame, cluster, size):
        out = np.empty((size,), np.int32)
        got = self._check(self.lib.fn["get_cluster_indices"](self._h, frame, cluster, out.ctypes.data, size),
                          "get_cluster_indices")
        return out[:got]

    def cluster_hull_candidates(self, frame, cluster, size):
        """Positions in the cluster's member list of the points that can be hull vertices (F2)."""
        out = np.empty(max(size, 1), np.int32)
        n = self._check(self.lib.fn["get_cluster_hull_candidates"](self._h, frame, cluster, out.ctypes.data, size),
                        "get_cluster_hull_candidates")
        return out[:min(n, size)]

    def max_plane(self, frame=0):
        m = HopeMaxPlane()
        self._check(self.lib.fn["get_max_plane"](self._h, frame, C.byref(m)), "get_max_plane")
        return m

    def max_plane_members(self, frame=0):
        """Every cluster of the RT winner (max_plane_cloud_), in point order; no limit on their number."""
        n = self._check(self.lib.fn["get_max_plane_members"](self._h, frame, None, 0), "get_max_plane_members")
        out = np.zeros(max(n, 1), np.int32)
        got = self._check(self.lib.fn["get_max_plane_members"](self._h, frame, out.ctypes.data, n),
                          "get_max_plane_members")
        assert got == n
        return [int(v) for v in out[:n]]

    def clusters_upon_plane(self, contour_xyz, frame=0, tolerance=0.01, min_size=10, max_size=240000):
        """Utilities::getClustersUponPlane on the last call's src_mono_cloud_: returns
        (n_upon, [(xyz[size,3], src_index[size]) per cluster, in the reference's order])."""
        contour = np.ascontiguousarray(contour_xyz, np.float32).reshape(-1, 3)
        n_cl, n_upon = C.c_int32(), C.c_int64()
        self._check(self.lib.fn["clusters_upon_plane"](self._h, frame, contour.ctypes.data_as(C.c_void_p), len(contour),
                                                       tolerance, min_size, max_size, C.byref(n_cl), C.byref(n_upon)),
                    "clusters_upon_plane")
        sizes = np.zeros(max(n_cl.value, 1), np.int32)
        got = self.lib.fn["get_upon_cluster_sizes"](self._h, sizes.ctypes.data_as(C.c_void_p), n_cl.value)
        assert got == n_cl.value
        out = []
        for k in range(n_cl.value):
            xyz = np.zeros((int(sizes[k]), 3), np.float32)
            src = np.zeros(int(sizes[k]), np.int32)
            r = self.lib.fn["get_upon_cluster"](self._h, k, xyz.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p), len(src))
            if r < 0:
                self._check(r, "get_upon_cluster")
            out.append((xyz, src))
        return n_upon.value, out

    def chain_sums(self, values):
        """Sequential float32 sums of the rows of `values` (chains x n) by exact_sum.hpp; returns (sums, stats)."""
        v = np.ascontiguousarray(values, dtype=np.float32)
        if v.ndim == 1:
            v = v[None, :]
        sums = np.zeros(v.shape[0], np.float32)
        stats = np.zeros(4, np.uint32)
        self._check(self.lib.fn["debug_chain_sums"](self._h, v.ctypes.data, v.shape[1], v.shape[0], sums.ctypes.data,
                                                    stats.ctypes.data), "debug_chain_sums")
        return sums, dict(zip(("slow_blocks", "float_steps", "plain_blocks", "top_hits"), (int(x) for x in stats)))

    def xs_stats(self):
        stats = np.zeros(4, np.uint32)
        self._check(self.lib.fn["debug_xs_stats"](self._h, stats.ctypes.data), "debug_xs_stats")
        return dict(zip(("slow_blocks", "float_steps", "plain_blocks", "top_hits"), (int(x) for x in stats)))

    def fast_division_active(self):
        return self.lib.fn["fast_division_active"](self._h)

    def profile_enable(self, on=True):
        self._check(self.lib.fn["profile_enable"](self._h, 1 if on else 0), "profile_enable")

    def profile(self):
        """[(stage name, total device ms, timed regions, algorithmic bytes)] accumulated since
        profile_enable()."""
        out = []
        for i in range(self.lib.fn["profile_stage_count"]()):
            name, ms, n, by = C.c_char_p(), C.c_double(), C.c_int64(), C.c_double()
            self.lib.fn["profile_stage"](self._h, i, C.byref(name), C.byref(ms), C.byref(n))
            self.lib.fn["profile_stage_bytes"](self._h, i, C.byref(by))
            out.append((name.value.decode(), ms.value, n.value, by.value))
        return out

    def timing(self):
        a, b = C.c_double(), C.c_double()
        self.lib.fn["last_timing"](self._h, C.byref(a), C.byref(b))
        return a.value, b.value
