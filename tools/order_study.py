#!/usr/bin/env python3
"""How much can the orders the oracle FIXES matter?  (CPU only; test infrastructure.)

PCL / FLANN / Eigen leave three orders unspecified that the oracle pins ("ORDER FIXED" in
oracle/hope_oracle.cpp): the order in which the points of a voxel are summed (VoxelGrid sorts with an
unstable std::sort on the voxel index only), the order of equidistant neighbours (FLANN: tree traversal),
and the summation order of normalAnalysis' 2xN * Nx2 covariance product (Eigen GEMM blocking).  Parity with
the real reference is unpinned there.  This script flips each choice to another order those libraries could
legally produce (oracle/Makefile `variants`) and reports, on the TUM example pair and the three golden
synthetic frames, what changes downstream: centroids (count, largest difference in ulp and in metres), the
normal-fit set, the clusters and the plane records (heights in mm; north_star's tolerance is 1 mm).

  make -C oracle variants && python tools/order_study.py            # prints the markdown table of DESIGN.md section 2
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hope_amd.capi import HopeContext, HopeLib  # noqa: E402

BUILD = os.path.join(ROOT, "oracle", "_build")
VARIANTS = [
    ("vsort", "voxel sums: literal `std::sort` on idx only (PCL's own call, libstdc++ 11)"),
    ("vdesc", "voxel sums: descending point index"),
    ("tiedesc", "equidistant neighbours: descending index"),
    ("covrev", "normalAnalysis covariance sums: reverse point order"),
]


def ulps(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def run(lib, inputs):
    out = []
    for depth, t, q in inputs:
        c = HopeContext(lib)
        c.set_pose(t, q)
        c.segment_depth(depth)
        cl = c.clusters(0)
        labels = c.labels(0)
        out.append({"vox": c.downsampled(0), "fit": c.fit_indices(0), "labels": labels, "clusters": cl,
                    "planes": cl[cl["pass"] == 1], "n_planes": c.counts(0).n_planes})
    return out


def compare(base, var):
    """One row of figures: base vs variant over all inputs."""
    n_vox = diff_vox = 0
    max_ulp = 0
    max_m = 0.0
    fit_sym = n_fit = 0
    lab_diff = lab_cmp = 0
    cl_diff = 0
    planes_b = planes_v = planes_unmatched = 0
    max_dz_mm = max_dbox_mm = 0.0
    for b, v in zip(base, var):
        same_vox = b["vox"].shape == v["vox"].shape
        if same_vox:
            u = ulps(b["vox"], v["vox"])
            n_vox += b["vox"].shape[0]
            diff_vox += int((u.max(axis=1) > 0).sum())
            max_ulp = max(max_ulp, int(u.max()))
            max_m = max(max_m, float(np.abs(b["vox"].astype(np.float64) - v["vox"].astype(np.float64)).max()))
        fb, fv = set(b["fit"].tolist()), set(v["fit"].tolist())
        fit_sym += len(fb ^ fv)
        n_fit += len(fb)
        if fb == fv:
            # clusters as sets of down-sampled points: does the partition change?
            lab_cmp += len(b["labels"])
            if len(b["clusters"]) == len(v["clusters"]) and np.array_equal(b["labels"], v["labels"]):
                pass
            else:
                lab_diff += int((b["labels"] != v["labels"]).sum()) if len(b["labels"]) == len(v["labels"]) else len(b["labels"])
        cl_diff += abs(len(b["clusters"]) - len(v["clusters"]))
        planes_b += len(b["planes"])
        planes_v += len(v["planes"])
        used = set()
        for p in b["planes"]:  # match every plane of the base by size + height
            best, bd = None, 1e9
            for j, r in enumerate(v["planes"]):
                if j in used:
                    continue
                d = abs(float(p["z_mean"]) - float(r["z_mean"])) + 1e-3 * abs(int(p["size"]) - int(r["size"]))
                if d < bd:
                    best, bd = j, d
            if best is None or abs(int(p["size"]) - int(v["planes"][best]["size"])) > max(3, 0.02 * int(p["size"])):
                planes_unmatched += 1
                continue
            used.add(best)
            r = v["planes"][best]
            max_dz_mm = max(max_dz_mm, 1e3 * abs(float(p["z_mean"]) - float(r["z_mean"])))
            for k in ("min_x", "min_y", "max_x", "max_y"):
                max_dbox_mm = max(max_dbox_mm, 1e3 * abs(float(p[k]) - float(r[k])))
    return {"vox": n_vox, "vox_diff": diff_vox, "max_ulp": max_ulp, "max_m": max_m, "fit": n_fit, "fit_sym": fit_sym,
            "lab_cmp": lab_cmp, "lab_diff": lab_diff, "cl_diff": cl_diff, "planes_b": planes_b, "planes_v": planes_v,
            "unmatched": planes_unmatched, "dz_mm": max_dz_mm, "dbox_mm": max_dbox_mm}


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "tum_example_depth.npz"))
    z = np.load(os.path.join(ROOT, "tests", "golden", "synth_frames_0_2.npz"))
    inputs = [(g["depth"], g["t"], g["q"])] + [(z["depth"][i], z["t"][i], z["q"][i]) for i in range(3)]
    base = run(HopeLib(os.path.join(BUILD, "libhope_oracle.so"), "hope_oracle_"), inputs)
    print("| order flipped | centroids that differ | largest difference | fit set (points in one set only) | "
          "labels that differ (same fit set) | clusters (count change) | planes base / variant (unmatched) | "
          "largest plane height difference | largest plane bbox difference |")
    print("|---|---|---|---|---|---|---|---|---|")
    for tag, what in VARIANTS:
        path = os.path.join(BUILD, f"libhope_oracle_{tag}.so")
        if not os.path.exists(path):
            print(f"| {what} | (not built: make -C oracle variants) |")
            continue
        r = compare(base, run(HopeLib(path, "hope_oracle_"), inputs))
        print(f"| {what} | {r['vox_diff']} of {r['vox']} | {r['max_ulp']} ulp, {r['max_m']:.2e} m | "
              f"{r['fit_sym']} of {r['fit']} | {r['lab_diff']} of {r['lab_cmp']} | {r['cl_diff']} | "
              f"{r['planes_b']} / {r['planes_v']} ({r['unmatched']}) | {r['dz_mm']:.4f} mm | {r['dbox_mm']:.4f} mm |")


if __name__ == "__main__":
    main()
