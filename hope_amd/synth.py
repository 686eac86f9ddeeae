"""Synthetic inputs of the shapes BASELINE.json / SURVEY.md §8d name (no datasets offline).

* make_depth_frames   — config 2: 640x480 uint16 TUM-style depth frames of a tabletop + floor
                        scene, per-frame camera pose as an un-normalised 4-decimal quaternion.
* make_loft_cloud     — config 3 stand-in: z-aligned multi-level "loft" cloud.
* make_multilevel_cloud — config 4: multi-level shelf scene, counter-based RNG keyed by the
                        global point index so that any sharding yields identical points.

Everything is torch code that runs on CPU or on the GPU (bench.py generates on the device).
These are test/bench inputs; they are not part of the extraction path.
"""
import math

import numpy as np
import torch

FX = FY = 517.0
CX, CY = 318.6, 255.3
DEPTH_SCALE = 5000.0


def _rot_to_quat(R):
    """3x3 rotation -> (x, y, z, w), Shepperd's method."""
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        w = 0.25 * s
        x = (R[2, 1] - R[1, 2]) / s
        y = (R[0, 2] - R[2, 0]) / s
        z = (R[1, 0] - R[0, 1]) / s
    elif R[0, 0] > R[1, 1] and R[0, 0] > R[2, 2]:
        s = math.sqrt(1.0 + R[0, 0] - R[1, 1] - R[2, 2]) * 2
        w = (R[2, 1] - R[1, 2]) / s
        x = 0.25 * s
        y = (R[0, 1] + R[1, 0]) / s
        z = (R[0, 2] + R[2, 0]) / s
    elif R[1, 1] > R[2, 2]:
        s = math.sqrt(1.0 + R[1, 1] - R[0, 0] - R[2, 2]) * 2
        w = (R[0, 2] - R[2, 0]) / s
        x = (R[0, 1] + R[1, 0]) / s
        y = 0.25 * s
        z = (R[1, 2] + R[2, 1]) / s
    else:
        s = math.sqrt(1.0 + R[2, 2] - R[0, 0] - R[1, 1]) * 2
        w = (R[1, 0] - R[0, 1]) / s
        x = (R[0, 2] + R[2, 0]) / s
        y = (R[1, 2] + R[2, 1]) / s
        z = 0.25 * s
    return np.array([x, y, z, w])


def _frame_params(frame):
    """Scene + camera of one frame (seed 1234 + frame)."""
    rs = np.random.RandomState(1234 + frame)
    pitch = math.radians(rs.uniform(35.0, 55.0))
    yaw = math.radians(rs.uniform(-10.0, 10.0))
    cam = np.array([0.0, 0.0, 1.4])
    # camera axes in the world (z up): x right, y down, z forward
    f = np.array([0.0, math.cos(pitch), -math.sin(pitch)])
    r = np.array([1.0, 0.0, 0.0])
    d = np.array([0.0, -math.sin(pitch), -math.cos(pitch)])
    cy_, sy_ = math.cos(yaw), math.sin(yaw)
    Rz = np.array([[cy_, -sy_, 0], [sy_, cy_, 0], [0, 0, 1.0]])
    R = Rz @ np.stack([r, d, f], axis=1)
    table_h = rs.uniform(0.70, 0.80)
    tc = Rz @ np.array([rs.uniform(-0.3, 0.3), rs.uniform(1.0, 1.5), 0.0])
    boxes = [(tc[0] - 0.6, tc[0] + 0.6, tc[1] - 0.4, tc[1] + 0.4, table_h - 0.04, table_h)]
    for _ in range(rs.randint(3, 7)):
        bw, bd, bh = rs.uniform(0.08, 0.25), rs.uniform(0.08, 0.25), rs.uniform(0.05, 0.30)
        bx = tc[0] + rs.uniform(-0.6 + bw / 2, 0.6 - bw / 2)
        by = tc[1] + rs.uniform(-0.4 + bd / 2, 0.4 - bd / 2)
        boxes.append((bx - bw / 2, bx + bw / 2, by - bd / 2, by + bd / 2, table_h, table_h + bh))
    while len(boxes) < 7:
        boxes.append((0, 0, 0, 0, -10.0, -10.0))  # degenerate, never hit
    q = np.round(_rot_to_quat(R), 4)
    t = np.round(cam, 4)
    return R, cam, np.array(boxes), t, q


def make_depth_frames(n_frames, first_frame=0, device="cpu", w=640, h=480, noise_seed=None):
    """Returns (depth uint16 [n, h, w], t float32 [n, 3], q_xyzw float32 [n, 4]) as torch tensors."""
    dev = torch.device(device)
    params = [_frame_params(first_frame + i) for i in range(n_frames)]
    R = torch.tensor(np.stack([p[0] for p in params]), dtype=torch.float32, device=dev)      # [n,3,3]
    cam = torch.tensor(np.stack([p[1] for p in params]), dtype=torch.float32, device=dev)    # [n,3]
    boxes = torch.tensor(np.stack([p[2] for p in params]), dtype=torch.float32, device=dev)  # [n,7,6]
    t = torch.tensor(np.stack([p[3] for p in params]), dtype=torch.float32)
    q = torch.tensor(np.stack([p[4] for p in params]), dtype=torch.float32)
    cols = torch.arange(w, device=dev, dtype=torch.float32)
    rows = torch.arange(h, device=dev, dtype=torch.float32)
    dx = ((cols - CX) / FX)[None, :].expand(h, w).reshape(-1)
    dy = ((rows - CY) / FY)[:, None].expand(h, w).reshape(-1)
    dcam = torch.stack([dx, dy, torch.ones_like(dx)], dim=0)  # [3, P]
    D = torch.matmul(R, dcam)                                 # [n, 3, P] world ray per unit camera z
    o = cam[:, :, None]
    inf = torch.tensor(float("inf"), device=dev)
    # floor z = 0
    tbest = torch.where(D[:, 2] < -1e-6, -o[:, 2] / D[:, 2], inf)
    for b in range(boxes.shape[1]):
        lo = boxes[:, b, 0::2][:, :, None]  # [n,3,1]
        hi = boxes[:, b, 1::2][:, :, None]
        inv = 1.0 / torch.where(D.abs() < 1e-9, torch.full_like(D, 1e-9), D)
        t0 = (lo - o) * inv
        t1 = (hi - o) * inv
        tnear = torch.minimum(t0, t1).amax(dim=1)
        tfar = torch.maximum(t0, t1).amin(dim=1)
        hit = (tnear <= tfar) & (tnear > 0)
        tbest = torch.where(hit & (tnear < tbest), tnear, tbest)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + first_frame if noise_seed is None else noise_seed)
    z = tbest + torch.randn(tbest.shape, generator=gen, device=dev) * (0.0015 * tbest * tbest)
    drop = torch.rand(tbest.shape, generator=gen, device=dev) < 0.05
    z = torch.where(torch.isfinite(z) & ~drop, z, torch.zeros_like(z))
    code = torch.clamp(torch.round(z * DEPTH_SCALE), 0, 65535).to(torch.int32)
    depth = code.to(torch.uint16) if hasattr(torch, "uint16") else code.to(torch.int16)
    return depth.reshape(n_frames, h, w), t, q


# ---------------------------------------------------------------------------------------------
def make_loft_cloud(n_points=5_000_000, seed=7):
    """Config-3 stand-in: z-aligned 12 m x 8 m x 3 m loft (floor, two mezzanines, stairs with
    0.18 m treads, walls), uniform area sampling, Gaussian noise sigma = 3 mm.  numpy float32 [n,3]."""
    rs = np.random.RandomState(seed)
    surfaces = []  # (kind, params, area); kind h: horizontal rect at z; v: vertical rect
    surfaces.append(("h", (0, 12, 0, 8, 0.0), 96.0))
    surfaces.append(("h", (0, 5, 0, 8, 1.2), 40.0))
    surfaces.append(("h", (8, 12, 0, 8, 2.4), 32.0))
    for s in range(6):  # stairs up to the first mezzanine
        surfaces.append(("h", (5 + 0.3 * s, 5.3 + 0.3 * s, 3, 4.2, 1.2 - 0.18 * (s + 1) + 0.0), 0.36))
    surfaces.append(("vx", (0.0, 0, 8, 0, 3), 24.0))   # wall at x = 0
    surfaces.append(("vx", (12.0, 0, 8, 0, 3), 24.0))
    surfaces.append(("vy", (0.0, 0, 12, 0, 3), 36.0))  # wall at y = 0
    surfaces.append(("vy", (8.0, 0, 12, 0, 3), 36.0))
    areas = np.array([s[2] for s in surfaces])
    counts = rs.multinomial(n_points, areas / areas.sum())
    parts = []
    for (kind, p, _), c in zip(surfaces, counts):
        u, v = rs.rand(c), rs.rand(c)
        if kind == "h":
            pts = np.stack([p[0] + u * (p[1] - p[0]), p[2] + v * (p[3] - p[2]), np.full(c, p[4])], 1)
        elif kind == "vx":
            pts = np.stack([np.full(c, p[0]), p[1] + u * (p[2] - p[1]), p[3] + v * (p[4] - p[3])], 1)
        else:
            pts = np.stack([p[1] + u * (p[2] - p[1]), np.full(c, p[0]), p[3] + v * (p[4] - p[3])], 1)
        parts.append(pts)
    xyz = np.concatenate(parts, 0)
    xyz += rs.randn(*xyz.shape) * 0.003
    xyz = xyz[rs.permutation(xyz.shape[0])]
    return xyz.astype(np.float32)


# ---------------------------------------------------------------------------------------------
def _hash_u32(idx, stream):
    """Counter-based RNG: 32 random bits per (index, stream) as int64 tensors (lowbias32 mix)."""
    m = 0xFFFFFFFF
    x = (idx * 0x9E3779B1 + (stream + 1) * 0x85EBCA77) & m
    x = (x ^ (x >> 16)) & m
    x = (x * 0x7FEB352D) & m
    x = (x ^ (x >> 15)) & m
    x = (x * 0x846CA68B) & m
    x = (x ^ (x >> 16)) & m
    x = (x + (idx >> 32) * 0x27D4EB2F) & m
    x = (x ^ (x >> 15)) & m
    x = (x * 0x2C1B3C6D) & m
    x = (x ^ (x >> 12)) & m
    return x


def _uniform(idx, stream):
    return (_hash_u32(idx, stream).to(torch.float64) + 0.5) / 4294967296.0


def _normal(idx, stream):
    u1, u2 = _uniform(idx, stream), _uniform(idx, stream + 1000)
    return torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)


def multilevel_layout(seed=42, extent=30.0, height=6.0):
    rs = np.random.RandomState(seed)
    levels = np.linspace(0.4, height - 0.6, 12)
    shelves = []
    for i in range(40):
        w_, d_ = rs.uniform(1.0, 4.0), rs.uniform(0.6, 2.5)
        x0, y0 = rs.uniform(0.5, extent - 0.5 - w_), rs.uniform(0.5, extent - 0.5 - d_)
        shelves.append((x0, x0 + w_, y0, y0 + d_, levels[i % 12]))
    return np.array(shelves)


def make_multilevel_cloud(n_points, first_index=0, seed=42, device="cpu", extent=30.0, height=6.0):
    """Config 4: points first_index .. first_index+n_points-1 of the multi-level scene
    (60 % horizontal shelves + floor, 30 % vertical panels, 10 % clutter, noise sigma = 2 mm).
    Returns a float32 torch tensor [n, 3] on `device`."""
    dev = torch.device(device)
    sh = torch.tensor(multilevel_layout(seed, extent, height), dtype=torch.float64, device=dev)
    idx = torch.arange(first_index, first_index + n_points, dtype=torch.int64, device=dev)
    base = seed * 16
    kind = _uniform(idx, base + 0)
    pick = (_uniform(idx, base + 1) * 41).to(torch.int64).clamp_(0, 40)  # 40 shelves + floor
    u, v, wz = _uniform(idx, base + 2), _uniform(idx, base + 3), _uniform(idx, base + 4)
    is_floor = pick == 40
    s = sh[pick.clamp(max=39)]
    hx = torch.where(is_floor, u * extent, s[:, 0] + u * (s[:, 1] - s[:, 0]))
    hy = torch.where(is_floor, v * extent, s[:, 2] + v * (s[:, 3] - s[:, 2]))
    hz = torch.where(is_floor, torch.zeros_like(u), s[:, 4])
    # vertical panel: the front face of the picked shelf, from the floor up to the shelf
    vx = torch.where(is_floor, torch.zeros_like(u), s[:, 0] + u * (s[:, 1] - s[:, 0]))
    vy = torch.where(is_floor, v * extent, s[:, 2])
    vz = torch.where(is_floor, wz * height, wz * s[:, 4])
    cx_, cy_, cz_ = u * extent, v * extent, wz * height
    horiz = kind < 0.6
    vert = (kind >= 0.6) & (kind < 0.9)
    x = torch.where(horiz, hx, torch.where(vert, vx, cx_))
    y = torch.where(horiz, hy, torch.where(vert, vy, cy_))
    z = torch.where(horiz, hz, torch.where(vert, vz, cz_))
    sig = 0.002
    x = x + sig * _normal(idx, base + 5)
    y = y + sig * _normal(idx, base + 6)
    z = z + sig * _normal(idx, base + 7)
    return torch.stack([x, y, z], dim=1).to(torch.float32).contiguous()
