#!/bin/bash
# usage (GPU box): tools/cloud_check.sh <tag> [points]
tag=${1:-tmp}; pts=${2:-100000000}
timeout -k 10 300 python tools/bench_cloud.py --kind multilevel --points $pts > gpurun_out/cloud_$tag.log 2>&1 || { tail -5 gpurun_out/cloud_$tag.log; exit 1; }
python3 - "$tag" <<'PY'
import json, sys
d = json.loads([x for x in open(f'gpurun_out/cloud_{sys.argv[1]}.log') if x.startswith('{')][-1])
print({k: v for k, v in d.items() if k != 'stages'})
for s in d['stages']:
    print(f"{s['stage']:28s} {s['ms']:8.3f} ms  {s['GBs']:8.1f} GB/s")
PY
