#!/bin/bash
# usage (GPU box): tools/ab_union_tiles.sh  -- the 1e8-point cloud step with k_union_local's tile at several sizes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
for t in ${TILES:-0 1 8192 12288}; do
  HOPE_DEBUG=1 HOPE_DEBUG_UNION_TILES=$t timeout -k 10 300 python bench.py --workload cloud --no-cpu-baseline > $O/cloud_ut$t.log 2>&1 || { tail -5 $O/cloud_ut$t.log; exit 1; }
  python - $O/cloud_ut$t.log $t <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
print('tile knob %s: %.2f ms  ' % (sys.argv[2], d['ms_per_step']) + '  '.join(f"{s['stage'].split('(')[0]}={s['ms_per_step']:.2f}" for s in d['stages']))
PY
done
