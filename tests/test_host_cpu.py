"""Host layer above the C ABI (hope_amd/host): format readers (F3), hull (F2), plane IDs (F4), the
PlaneSegment / PlaneSegmentRT adapters and the hope_node executable.  CPU only: the adapters are linked on
the oracle by tests/Makefile (test-only build); the product build links libhope_hip.so and is exercised by
tests/test_gpu_parity.py on the GPU box."""
import ctypes
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

from hope_amd.capi import HopeContext, HOPE_MODE_RT

HOST_LIB = os.path.join(ROOT, "hope_amd", "host", "libhope_host.so")
NODE_ORACLE = os.path.join(ROOT, "tests", "_build", "hope_node_oracle")
RT_ORACLE = os.path.join(ROOT, "tests", "_build", "rt_driver_oracle")


@pytest.fixture(scope="session")
def host():
    lib = ctypes.CDLL(HOST_LIB)
    lib.hope_host_last_error.restype = ctypes.c_char_p
    lib.hope_host_read_cloud.restype = ctypes.c_longlong
    return lib


@pytest.fixture(scope="session")
def test_bins(oracle_lib):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests"), "_build/hope_node_oracle", "_build/rt_driver_oracle"])
    return NODE_ORACLE, RT_ORACLE


def read_png(host, path):
    w, h, ch, bd = (ctypes.c_int() for _ in range(4))
    assert host.hope_host_read_png(path.encode(), None, ctypes.c_size_t(0), ctypes.byref(w), ctypes.byref(h), ctypes.byref(ch),
                                   ctypes.byref(bd)) == 0, host.hope_host_last_error()
    out = np.zeros((h.value, w.value, ch.value), np.uint16 if bd.value == 16 else np.uint8)
    assert host.hope_host_read_png(path.encode(), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(out.nbytes), ctypes.byref(w),
                                   ctypes.byref(h), ctypes.byref(ch), ctypes.byref(bd)) == 0
    return out


def png_bytes(arr, ctype, depth, filters=None, palette=None):
    """Minimal PNG writer with a chosen filter type per row (PIL picks its own, so write by hand)."""
    h, w = arr.shape[:2]
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    a = arr.reshape(h, w * ch)
    rows = a.astype(">u2").view(np.uint8).reshape(h, -1) if depth == 16 else a.astype(np.uint8)
    bpp = max(1, ch * depth // 8)
    raw = bytearray()
    prev = np.zeros(rows.shape[1], np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        ft = filters[y % len(filters)] if filters else 0
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0: enc = cur
        elif ft == 1: enc = cur - left
        elif ft == 2: enc = cur - prev
        elif ft == 3: enc = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            enc = cur - pred
        raw.append(ft)
        raw += bytes((enc & 255).astype(np.uint8))
        prev = cur
    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body))
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    z = zlib.compress(bytes(raw))
    out += chunk(b"IDAT", z[: len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:]) + chunk(b"IEND", b"")
    return out


def test_png_reader_depth_and_rgb(host, tmp_path, tum_example):
    depth = tum_example[0]
    p = tmp_path / "d.png"
    p.write_bytes(png_bytes(depth, 0, 16, filters=[0, 1, 2, 3, 4]))
    assert np.array_equal(read_png(host, str(p))[:, :, 0], depth)
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    p.write_bytes(png_bytes(rgb, 2, 8, filters=[4, 3, 1]))
    assert np.array_equal(read_png(host, str(p)), rgb)
    rgba = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    p.write_bytes(png_bytes(rgba, 6, 8, filters=[2]))
    assert np.array_equal(read_png(host, str(p)), rgba)
    # the same files through PIL's encoder (its own filter choice and IDAT split)
    from PIL import Image
    Image.fromarray(depth).save(p)
    assert np.array_equal(read_png(host, str(p))[:, :, 0], depth)
    Image.fromarray(rgb).save(p)
    assert np.array_equal(read_png(host, str(p)), rgb)
    pal = Image.fromarray(rgb).convert("P", palette=Image.ADAPTIVE, colors=16)
    pal.save(p)
    assert np.array_equal(read_png(host, str(p)), np.asarray(pal.convert("RGB")))
    p.write_bytes(b"not a png at all")
    w = ctypes.c_int()
    assert host.hope_host_read_png(str(p).encode(), None, ctypes.c_size_t(0), *(ctypes.byref(w),) * 4) == -1
    assert b"PNG" in host.hope_host_last_error()


def read_cloud(host, path, kind, n):
    xyz = np.zeros((n, 3), np.float32)
    rgb = np.zeros((n, 3), np.uint8)
    got = host.hope_host_read_cloud(str(path).encode(), kind.encode(), xyz.ctypes.data_as(ctypes.c_void_p),
                                    rgb.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n))
    assert got == n, host.hope_host_last_error()
    return xyz, rgb


def test_ply_and_pcd_readers(host, tmp_path):
    rng = np.random.default_rng(5)
    n = 257
    xyz = rng.normal(size=(n, 3)).astype(np.float32)
    rgb = rng.integers(0, 256, (n, 3), dtype=np.uint8)
    hdr = "ply\nformat {}\ncomment made by a test\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" \
          "property float nx\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nelement face 1\n" \
          "property list uchar int vertex_indices\nend_header\n" % n
    for fmt, e in (("binary_little_endian 1.0", "<"), ("binary_big_endian 1.0", ">")):
        body = b"".join(struct.pack(e + "ffffBBB", *xyz[i], 0.5, *rgb[i]) for i in range(n)) + struct.pack(e + "Biii", 3, 0, 1, 2)
        p = tmp_path / "c.ply"
        p.write_bytes(hdr.format(fmt).encode() + body)
        x, c = read_cloud(host, p, "ply", n)
        assert np.array_equal(x, xyz) and np.array_equal(c, rgb)
    p.write_text(hdr.format("ascii 1.0") + "".join("%r %r %r 0.5 %d %d %d\n" % (*map(float, xyz[i]), *rgb[i]) for i in range(n)) + "3 0 1 2\n")
    x, c = read_cloud(host, p, "ply", n)
    assert np.array_equal(x, xyz) and np.array_equal(c, rgb)
    # PCD: ascii, binary, binary_compressed (LZF stream of literals with one hand-made back reference)
    packed = (rgb[:, 0].astype(np.uint32) << 16 | rgb[:, 1].astype(np.uint32) << 8 | rgb[:, 2]).astype(np.uint32)
    xyz[3] = np.nan
    head = "# .PCD v0.7\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F {}\nCOUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA {}\n" % (n, n)
    q = tmp_path / "c.pcd"
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgb", "<u4")])
    rec["x"], rec["y"], rec["z"], rec["rgb"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], packed
    q.write_bytes(head.format("U", "binary").encode() + rec.tobytes())
    x, c = read_cloud(host, q, "pcd", n)
    assert np.array_equal(x, xyz, equal_nan=True) and np.array_equal(c, rgb)
    q.write_text(head.format("U", "ascii") + "".join("%s %s %s %d\n" % (*("nan" if np.isnan(v) else repr(float(v)) for v in xyz[i]), packed[i]) for i in range(n)))
    x, c = read_cloud(host, q, "pcd", n)
    assert np.array_equal(x, xyz, equal_nan=True) and np.array_equal(c, rgb)
    soa = b"".join(np.ascontiguousarray(rec[k]).tobytes() for k in ("x", "y", "z", "rgb"))   # one array per field
    comp = bytearray()
    for i in range(0, len(soa), 32):                  # LZF made of literal runs only (ctrl = length - 1 < 32)
        lit = soa[i:i + 32]
        comp.append(len(lit) - 1)
        comp += lit
    q.write_bytes(head.format("U", "binary_compressed").encode() + struct.pack("<II", len(comp), len(soa)) + bytes(comp))
    x, c = read_cloud(host, q, "pcd", n)
    assert np.array_equal(x, xyz, equal_nan=True) and np.array_equal(c, rgb)
    # a stream with a real back reference: 8 literal bytes then "copy 8 bytes from 8 back"
    lit = struct.pack("<f", 1.5)
    comp = bytes([3]) + lit + bytes([(7 << 5) | 0, 12 - 2 - 7, 3])      # then 12 more bytes from offset 4 back (encoded 3)
    q.write_bytes(head.format("U", "binary_compressed").replace("WIDTH %d" % n, "WIDTH 1").replace("POINTS %d" % n, "POINTS 1").encode()
                  + struct.pack("<II", len(comp), 16) + comp)
    x, c = read_cloud(host, q, "pcd", 1)
    assert np.array_equal(x, [[1.5, 1.5, 1.5]]) and list(c[0]) == [0xC0, 0x00, 0x00]
    assert host.hope_host_read_cloud(str(q).encode(), b"obj", None, None, ctypes.c_size_t(0)) == -1


def test_tum_list_reader(host, tmp_path):
    p = tmp_path / "all.txt"
    p.write_text("# stamp rgb depth pose\n"
                 "1305031453.359684 rgb/1305031453.359684.png depth/1305031453.374112.png 1.3405 0.6266 1.6575 0.6574 0.6126 -0.2949 -0.3248\n"
                 "broken line\n"
                 "1305031453.391690 rgb/1305031453.391690.png depth/1305031453.404816.png 1.3303 0.6256 1.6464 0.6579 0.6161 -0.2932 -0.3189\n")
    pose = np.zeros((4, 7), np.float32)
    assert host.hope_host_read_tum_list(str(p).encode(), pose.ctypes.data_as(ctypes.c_void_p), 4) == 2
    assert np.allclose(pose[1], [1.3303, 0.6256, 1.6464, 0.6579, 0.6161, -0.2932, -0.3189])


def hull(host, xyz):
    xyz = np.ascontiguousarray(xyz, np.float32)
    idx = np.zeros(len(xyz), np.int32)
    area = ctypes.c_double()
    n = host.hope_host_hull(xyz.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(len(xyz)), ctypes.c_size_t(12),
                            idx.ctypes.data_as(ctypes.c_void_p), len(xyz), ctypes.byref(area))
    return idx[:n], area.value


def test_hull_against_qhull(host):
    """scipy.spatial.ConvexHull IS qhull (the library pcl::ConvexHull calls): same vertex set and area
    (qhull's `volume` in 2-D), and PCL's output order (ascending atan2 about the vertex centroid)."""
    from scipy.spatial import ConvexHull
    rng = np.random.default_rng(11)
    for n in (3, 10, 500, 20000):
        xyz = np.concatenate([rng.uniform(-2, 2, (n, 2)), rng.normal(0.8, 0.002, (n, 1))], 1).astype(np.float32)
        idx, area = hull(host, xyz)
        ref = ConvexHull(xyz[:, :2].astype(np.float64))
        assert sorted(idx) == sorted(ref.vertices)
        assert abs(area - ref.volume) <= 1e-9 * max(1.0, ref.volume)
        c = xyz[np.sort(idx), :2].mean(0, dtype=np.float32)
        ang = np.arctan2(xyz[idx, 1] - c[1], xyz[idx, 0] - c[0])
        assert np.all(np.diff(ang) > 0)
    # collinear and duplicate points are not vertices; a degenerate patch gives what is there
    sq = np.array([[0, 0, 1], [1, 0, 1], [2, 0, 1], [2, 2, 1], [0, 2, 1], [1, 1, 1], [0, 0, 1], [1, 2, 1]], np.float32)
    idx, area = hull(host, sq)
    assert sorted(idx) == [0, 2, 3, 4] and area == 4.0
    idx, area = hull(host, sq[:2])
    assert sorted(idx) == [0, 1] and area == 0.0


def match_id(host, g, l, ids):
    g, l = np.ascontiguousarray(g, np.float32), np.ascontiguousarray(l, np.float32)
    ids = np.ascontiguousarray(ids, np.int32)
    out = np.zeros(len(l), np.int32)
    host.hope_host_match_id(g.ctypes.data_as(ctypes.c_void_p), len(g), l.ctypes.data_as(ctypes.c_void_p), len(l), 5,
                            ids.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return list(out)


def test_plane_id_matching(host):
    g = np.array([[0.0, -1, -1, 1, 1], [0.75, 0, 0, 1.2, 0.8], [1.1, 0.2, 0.2, 0.5, 0.5]], np.float32)
    assert match_id(host, g, g[[2, 0, 1]] + 0.001, [4, 7, 9]) == [9, 4, 7]
    # a new plane takes the smallest id not in use; two locals closest to one global: the nearer keeps the id
    l = np.array([[0.76, 0, 0, 1.2, 0.8], [0.80, 0.1, 0.1, 1.0, 0.7], [0.0, -1, -1, 1, 1]], np.float32)
    assert match_id(host, g[:2], l, [0, 1]) == [1, 2, 0]
    # one global and one identical local: every column has sd 0, the z-scores are NaN, no distance compares
    # below 10000, so the plane gets a fresh id -- the reference's arithmetic (utilities.cpp:959-969,1112)
    assert match_id(host, g[:1], g[:1], [1023]) == [0]


def test_pointcloud2_unpacking(host):
    rng = np.random.default_rng(2)
    w, h, step = 7, 3, 32
    xyz = rng.normal(size=(h * w, 3)).astype(np.float32)
    for big in (0, 1):
        buf = np.zeros((h, w * step + 5), np.uint8)      # row padding
        for i, p in enumerate(xyz):
            r, c = divmod(i, w)
            buf[r, c * step:c * step + 4] = np.frombuffer(struct.pack(">f" if big else "<f", p[0]), np.uint8)
            buf[r, c * step + 4:c * step + 8] = np.frombuffer(struct.pack(">f" if big else "<f", p[1]), np.uint8)
            buf[r, c * step + 16:c * step + 24] = np.frombuffer(struct.pack(">d" if big else "<d", float(p[2])), np.uint8)
        names = (ctypes.c_char_p * 4)(b"x", b"y", b"rgb", b"z")
        offs = (ctypes.c_uint * 4)(0, 4, 8, 16)
        types = (ctypes.c_ubyte * 4)(7, 7, 7, 8)
        out = np.zeros((h * w, 3), np.float32)
        assert host.hope_host_from_pointcloud2(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes), w, h, step, w * step + 5, 4,
                                               names, offs, types, big, out.ctypes.data_as(ctypes.c_void_p)) == 0, host.hope_host_last_error()
        assert np.array_equal(out, xyz)


def run_node(binary, args, tmp_path, hull=True):
    dump = tmp_path / ("dump_%s.jsonl" % os.path.basename(binary))
    if dump.exists():
        dump.unlink()
    env = dict(os.environ, HOPE_NODE_DUMP=str(dump))
    if hull:
        env["HOPE_NODE_HULL"] = "1"
    r = subprocess.run([binary] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=600)
    recs = [json.loads(l) for l in dump.read_text().splitlines()] if dump.exists() else []
    return r, recs


def write_tum_dataset(tmp_path, depth_frames, t, q):
    from PIL import Image
    d = tmp_path / "tum"
    (d / "depth").mkdir(parents=True)
    (d / "rgb").mkdir()
    lines = []
    for i, fr in enumerate(depth_frames):
        Image.fromarray(fr).save(d / "depth" / ("%d.png" % i))
        Image.fromarray(np.zeros((fr.shape[0], fr.shape[1], 3), np.uint8)).save(d / "rgb" / ("%d.png" % i))
        lines.append("%d.5 rgb/%d.png depth/%d.png %s %s" % (i, i, i, " ".join(repr(float(v)) for v in t[i]), " ".join(repr(float(v)) for v in q[i])))
    (d / "all.txt").write_text("\n".join(lines) + "\n")
    return d


def expected_planes(lib, depth, t, q):
    c = HopeContext(lib)
    c.set_pose(t, q)
    c.segment_depth(depth)
    k = c.counts(0)
    cl = c.clusters(0)
    return k, [r for r in cl if r["pass"] == 1]


def check_record(rec, k, planes):
    assert (rec["n_raw"], rec["n_vox"], rec["n_fit"], rec["n_clusters"]) == (k.n_raw, k.n_vox, k.n_fit, k.n_clusters)
    assert len(rec["planes"]) == len(planes)
    for got, want in zip(rec["planes"], planes):
        assert np.array_equal(np.array(got["coeff"], np.float32),
                              np.array([want["z_mean"], want["min_x"], want["min_y"], want["max_x"], want["max_y"]], np.float32))
        assert got["points"] == want["size"]


def test_hope_node_single_tum_pair(test_bins, oracle_lib, tum_example, tmp_path):
    """`hope_node {prefix} {rgb} {depth} {tx ty tz qx qy qz qw}` (README.md:51) == the C ABI called directly."""
    depth, t, q = tum_example
    d = write_tum_dataset(tmp_path, [depth], [t], [q])
    args = [str(d) + "/", "rgb/0.png", "depth/0.png"] + ["%.4f" % v for v in t] + ["%.4f" % v for v in q]
    r, recs = run_node(test_bins[0], args, tmp_path)
    assert r.returncode == 0, r.stderr
    assert "Using single pair of rgb and depth image" in r.stdout and "Point number after down sampling: #5699" in r.stdout
    k, planes = expected_planes(oracle_lib, depth, np.float32([float("%.4f" % v) for v in t]), np.float32([float("%.4f" % v) for v in q]))
    assert len(recs) == 1
    check_record(recs[0], k, planes)
    assert [p["id"] for p in recs[0]["planes"]] == list(range(len(planes)))        # first detection: ids 0..n-1
    assert all(p["hull"] >= 3 and p["area"] > 0 for p in recs[0]["planes"])
    # the desk top of the example is about 1.2 m^2 of visible surface
    big = max(recs[0]["planes"], key=lambda p: p["points"])
    assert 0.3 < big["area"] < 3.0


def test_hope_node_list_mode_and_argument_counts(test_bins, oracle_lib, tmp_path):
    g = np.load(os.path.join(GOLDEN, "synth_frames_0_2.npz"))
    d = write_tum_dataset(tmp_path, g["depth"], g["t"], g["q"])
    r, recs = run_node(test_bins[0], [d], tmp_path, hull=False)
    assert r.returncode == 0, r.stderr
    assert len(recs) == 3 and r.stdout.count("Processing: rgb/") == 3
    ids_seen = []
    for i, rec in enumerate(recs):
        # phaseInput keeps the LAST line's translation for every frame (hope_node.cpp:88-90,231)
        k, planes = expected_planes(oracle_lib, g["depth"][i], g["t"][2], g["q"][i])
        check_record(rec, k, planes)
        ids_seen.append([p["id"] for p in rec["planes"]])
    assert all(len(set(ids)) == len(ids) for ids in ids_seen)                       # ids unique within a frame
    # wrong argument counts print the help and return -1; three numbers select the (ROS-only) synthetic mode
    r, _ = run_node(test_bins[0], ["a", "b", "c", "d", "e"], tmp_path)
    assert r.returncode == 255 and "hope_node VARIABLES" in r.stdout
    r, _ = run_node(test_bins[0], ["0", "0", "0"], tmp_path)
    assert r.returncode == 2 and "Using synthetic object." in r.stdout and "no source cloud" in r.stderr
    r, _ = run_node(test_bins[0], [tmp_path / "none.xyz", "xyz"], tmp_path)
    assert r.returncode == 255 and "Unrecognized file format." in r.stderr


def test_hope_node_point_cloud_file(test_bins, oracle_lib, tmp_path):
    xyz = np.load(os.path.join(GOLDEN, "loft_120k.npz"))["xyz"]
    p = tmp_path / "loft.ply"
    p.write_bytes(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                   "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(xyz)).encode()
                  + np.concatenate([xyz.view(np.uint8).reshape(len(xyz), 12), np.zeros((len(xyz), 3), np.uint8)], 1).tobytes())
    r, recs = run_node(test_bins[0], [p, "ply"], tmp_path, hull=False)
    assert r.returncode == 0, r.stderr
    c = HopeContext(oracle_lib)
    c.segment_cloud(xyz, apply_filter_transform=False)
    check_record(recs[0], c.counts(0), [r_ for r_ in c.clusters(0) if r_["pass"] == 1])


def test_hope_node_synthetic_mode_with_rpy(test_bins, oracle_lib, tmp_path, monkeypatch):
    """`hope_node roll pitch yaw` (hope_node.cpp:129-135,169): the cloud of the (ROS) callback comes from a file; it is
    range-filtered and transformed with the setRPY pose (plane_segment.cpp:617-621, transform.cpp:108-150)."""
    from test_oracle import rpy_scene
    rpy = (0.3, -0.2, 0.5)
    xyz = rpy_scene(*rpy, n=40000)
    p = tmp_path / "syn.ply"
    p.write_bytes(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                   "end_header\n" % len(xyz)).encode() + xyz.tobytes())
    monkeypatch.setenv("HOPE_NODE_SYN_CLOUD", str(p))
    r, recs = run_node(test_bins[0], [repr(v) for v in rpy], tmp_path, hull=False)
    assert r.returncode == 0, r.stderr
    c = HopeContext(oracle_lib)
    c.set_pose_rpy(*rpy)
    c.segment_cloud(xyz, apply_filter_transform=True)
    planes = [r_ for r_ in c.clusters(0) if r_["pass"] == 1]
    assert len(planes) >= 3
    check_record(recs[0], c.counts(0), planes)


def test_rt_adapter(test_bins, oracle_lib, tum_example, tmp_path):
    """PlaneSegmentRT: cloudCallback + configCallback + getHorizontalPlanes == the RT ctx called directly; the
    contour is the hull of the winner's points."""
    depth, t, q = tum_example
    z = depth.astype(np.float32) / np.float32(5000.0)
    cols = (np.arange(640, dtype=np.float32) - np.float32(318.6))[None, :]
    rows = (np.arange(480, dtype=np.float32) - np.float32(255.3))[:, None]
    cloud = np.stack([cols * z / np.float32(517.0), rows * z / np.float32(517.0), z], -1).reshape(-1, 3)
    cloud[(z.reshape(-1) <= 0.3) | (z.reshape(-1) >= 8.0)] = np.nan
    f = tmp_path / "cloud.f32"
    cloud.astype(np.float32).tofile(f)
    for lo, hi, aggr in ((0.5, 1.5, 1), (0.5, 1.5, 0), (-0.2, 0.3, 1), (3.0, 4.0, 1)):
        r = subprocess.run([test_bins[1], str(f)] + [repr(float(v)) for v in t] + [repr(float(v)) for v in q] + [str(lo), str(hi), str(aggr)],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        got = json.loads(r.stdout.strip().splitlines()[-1])
        c = HopeContext(oracle_lib, mode=HOPE_MODE_RT, min_height=lo, max_height=hi, aggressive_merge=aggr)
        c.set_pose(t, q)
        c.segment_cloud(cloud, apply_filter_transform=True)
        m = c.max_plane(0)
        if not m.valid:
            assert got["points"] == 0 and got["contour"] == [] and "No plane detected" in r.stdout
            continue
        assert got["points"] == m.n_points and np.float32(got["z"]) == np.float32(m.z)
        vox, fit = c.downsampled(0), c.fit_indices(0)
        cl = c.clusters(0)
        pts = np.concatenate([vox[fit[c.cluster_indices(0, ci, int(cl["size"][ci]))]] for ci in c.max_plane_members(0)])
        from scipy.spatial import ConvexHull
        ref = ConvexHull(pts[:, :2].astype(np.float64))
        contour = np.array(got["contour"], np.float32)
        assert sorted(map(tuple, contour)) == sorted(map(tuple, pts[ref.vertices]))
        assert abs(got["area"] - ref.volume) < 1e-6


def test_rt_adapter_merges_every_cluster_of_one_height(test_bins, oracle_lib, tmp_path):
    """max_plane_cloud_ holds the points of ALL merged clusters (plane_segment.cpp:943-960), also beyond
    sixteen of them: its size is the winner's point count."""
    from conftest import patch_cloud
    xyz = patch_cloud(24)
    f = tmp_path / "patches.f32"
    xyz.tofile(f)
    r = subprocess.run([test_bins[1], str(f), "0", "0", "0", "0", "0", "0", "1", "0.8", "1.5", "1"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    c = HopeContext(oracle_lib, mode=HOPE_MODE_RT, min_height=0.8, max_height=1.5, aggressive_merge=1)
    c.segment_cloud(xyz, apply_filter_transform=True)
    m = c.max_plane(0)
    assert m.n_members >= 20 and got["points"] == m.n_points
