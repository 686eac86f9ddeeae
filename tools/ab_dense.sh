#!/bin/bash
# usage (GPU box): tools/ab_dense.sh  -- kernel times of the frame bench with dense and with sparse group records
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for d in 1 0; do
  HOPE_DEBUG=1 HOPE_DEBUG_DENSE_RECORDS=$d timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dense$d -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cloud --no-stream --no-two-ranges > $R/gpurun_out/prof_dense$d.log 2>&1 || { tail -5 $R/gpurun_out/prof_dense$d.log; exit 1; }
  python3 - $R/gpurun_out/prof_dense$d <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in ('k_bbox_records', 'k_voxel_rank', 'k_centroids_lists', 'k_group_keys', 'k_frame_vox_scan', 'k_scan_')):
        print(sys.argv[1][-1], n.replace('void hope::', '')[:50], r['Calls'], '%.1f us' % (float(r['AverageNs']) / 1e3))
PY
done
