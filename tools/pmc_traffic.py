"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, each collected
in its own run as /opt/skills/guides/MI355X_MICROARCH.md prescribes):

   python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write [steps+warmup] > profiles/rNN_hbm_traffic.md
   (also writes profiles/hbm_traffic.json: stage -> HBM bytes per bench step, read by bench.py)

Units: FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950 correction: FETCH_SIZE counts 64 B per 128-B
request of a wide coalesced read, i.e. exactly half the bytes of such a stream (the guide's
calibration); both the raw and the doubled figure are printed, the doubled one is what
hbm_traffic.json records (an upper bound for the gather-type kernels, whose narrower requests
are not calibrated)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

STAGE_OF = [  # kernel name fragment -> bench.py stage
    ("k_part_count", "k_part_count"), ("k_part_scatter", "k_part_scatter"), ("k_part_", "k_part_finish"),
    ("k_grid_", "column_index"), ("k_xs_", "k_cluster_verify"), ("k_big_", "k_cluster_verify"),
    ("k_bbox", "k_bbox"), ("k_frame_grid", "k_frame_grid"), ("k_group_records", "k_voxel_keys"),
    ("k_group_keys", "k_voxel_keys"), ("k_fit_sorted", "fit_column_index"), ("k_fit_col_table", "fit_column_index"),
    ("k_voxel_rank", "k_voxel_rank"), ("k_frame_vox_scan", "k_voxel_rank"), ("k_voxel_keys", "k_voxel_keys"), ("k_centroids", "k_centroids"), ("k_column_keys", "k_column_keys"), ("k_column_index", "column_index"),
    ("k_column_table", "column_index"), ("k_gather_sorted", "column_index"),
    ("k_normals", "k_normals"), ("k_fit_", "fit_gate+compact"), ("k_frame_fit", "fit_gate+compact"),
    ("k_knn_todo", "k_knn_todo"), ("k_knn", "k_knn"), ("k_coarse", "k_knn_todo"), ("k_gather_coarse", "k_knn_todo"),
    ("k_seed_ranks", "labels(union+sweeps)"), ("k_union", "labels(union+sweeps)"),
    ("k_flatten_roots", "labels(union+sweeps)"), ("k_cross_edges", "labels(union+sweeps)"),
    ("k_label_", "labels(union+sweeps)"), ("k_segment_", "segments(scan+sort)"),
    ("k_frame_cluster", "segments(scan+sort)"), ("k_point_clusters", "segments(scan+sort)"),
    ("k_cluster_heads", "segments(scan+sort)"), ("k_members_by_frame", "segments(scan+sort)"),
    ("k_cluster_verify", "k_cluster_verify"),
]


def load(d, counter):
    tot, calls = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            tot[r["Kernel_Name"]] += float(r["Counter_Value"]) * 1024.0
            calls[r["Kernel_Name"]] += 1
    return tot, calls


def main():
    """pmc_traffic.py <fetch dir> <write dir> [steps] [output json name] [kernel filter: frames|cloud]"""
    fetch, calls = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    out_name = sys.argv[4] if len(sys.argv) > 4 else "hbm_traffic.json"
    rows = []
    for k in set(fetch) | set(write):
        if "hope::" not in k and "rocprim" not in k:
            continue
        rows.append((fetch.get(k, 0.0) + write.get(k, 0.0), k))
    stage = defaultdict(float)
    print("| kernel | launches | FETCH_SIZE MB/launch (raw) | x2 (gfx950) | WRITE_SIZE MB/launch |")
    print("|---|---|---|---|---|")
    for _, k in sorted(rows, reverse=True):
        n = max(calls.get(k, 1), 1)
        short = k.replace("void ", "").replace("hope::", "").split("(")[0][:60]
        print(f"| `{short}` | {n} | {fetch.get(k, 0) / n / 1e6:.1f} | {2 * fetch.get(k, 0) / n / 1e6:.1f} | "
              f"{write.get(k, 0) / n / 1e6:.1f} |")
        for frag, st in STAGE_OF:
            if frag in k:
                stage[st] += (2 * fetch.get(k, 0) + write.get(k, 0)) / steps
                break
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "profiles", out_name)
    json.dump({k: round(v) for k, v in stage.items()}, open(out, "w"), indent=1, sort_keys=True)
    # where the figures come from: bench.py copies this next to roofline.traffic
    import subprocess
    try:
        head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], text=True).strip()
        dirty = bool(subprocess.check_output(["git", "-C", root, "status", "--porcelain", "--", "hope_amd", "bench.py"], text=True).strip())
    except Exception:
        head, dirty = os.environ.get("HOPE_PROFILED_HEAD", "unknown"), False
    json.dump({"profiled_build": head + ("+uncommitted" if dirty else ""), "counters": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one "
               "pass each; FETCH_SIZE doubled (gfx950)", "steps_profiled": steps},
              open(out.replace(".json", ".meta.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
