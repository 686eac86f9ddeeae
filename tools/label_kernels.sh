#!/bin/bash
# usage (GPU box): tools/label_kernels.sh <tag>  -- per-kernel times of the labels / segments stages, frame batch and 1e8-point cloud
tag=${1:-lk}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in frames cloud; do
  rm -rf $O/prof_lk_$w
  if [ $w = frames ]; then A="--steps 4 --warmup 2 --no-cpu-baseline --no-cloud --no-stream --no-two-ranges"; else A="--workload cloud --no-cpu-baseline"; fi; [ -n "$HIPLIB" ] && A="$A --hip-lib $R/$HIPLIB"
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lk_$w -- python3 $R/bench.py $A > $O/lk_${tag}_$w.log 2>&1 || { tail -5 $O/lk_${tag}_$w.log; exit 1; }
  python3 - $O/prof_lk_$w $w <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
keys = ('k_union', 'k_flatten', 'k_cross_edges', 'k_label', 'k_segment', 'k_seed_ranks', 'k_point_clusters', 'k_cluster_heads', 'k_members', 'k_cluster_order')
print(sys.argv[2])
for r in csv.DictReader(open(f)):
    n = r['Name']
    if any(k in n for k in keys):
        print('  %-48s %4s calls  %8.1f us' % (n.replace('void hope::', '').replace('hope::', '')[:48], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
