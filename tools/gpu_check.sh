#!/bin/bash
# usage: tools/gpu_check.sh [tag]  -- on the GPU box: parity tests, then one bench line (no CPU baseline)
tag=${1:-tmp}
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_$tag.log 2>&1
rc=$?
tail -3 gpurun_out/pytest_$tag.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_$tag.log 2>&1 || { tail -5 gpurun_out/bench_$tag.log; exit 1; }
python3 - "$tag" <<'PY'
import json, sys
d = json.loads([x for x in open(f'gpurun_out/bench_{sys.argv[1]}.log') if x.startswith('{')][-1])
print('value %.4g pts/s  %.3f ms/step' % (d['value'], d['ms_per_step']))
for s in d['stages']:
    print(f"{s['stage']:28s} {s['ms_per_step']:8.3f} ms  {s['achieved_GBs']:8.1f} GB/s")
PY
