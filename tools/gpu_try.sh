#!/bin/bash
# usage (GPU box): tools/gpu_try.sh <tag> [pytest -k filter]  -- GPU parity tests, then the bench with and without sub-batches
tag=${1:-try}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ -n "$2" ]; then K=(-k "$2"); else K=(); fi
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 150 --timeout-method thread "${K[@]}" > $O/pytest_$tag.log 2>&1
rc=$?
tail -5 $O/pytest_$tag.log
[ $rc -ne 0 ] && exit $rc
for sb in ${SBS:-1}; do
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cloud --no-stream --no-two-ranges --sub-batches $sb > $O/bench_${tag}_sb$sb.log 2>&1 || { tail -5 $O/bench_${tag}_sb$sb.log; exit 1; }
  python - $O/bench_${tag}_sb$sb.log <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
print(sys.argv[1].split('/')[-1], 'value %.4g pts/s  %.3f ms/step' % (d['value'], d['ms_per_step']))
print('  ' + '  '.join(f"{s['stage'].split('(')[0]}={s['ms_per_step']:.2f}" for s in d['stages']))
PY
done
if [ -n "$ALL" ]; then  # the other workloads too: a change that helps the frame batch can hurt these
  timeout -k 10 300 python bench.py --workload cloud --no-cpu-baseline > $O/cloud_$tag.log 2>&1 || { tail -5 $O/cloud_$tag.log; exit 1; }
  timeout -k 10 200 python tools/bench_rt.py > $O/rt_$tag.json 2> $O/rt_$tag.err || { tail -5 $O/rt_$tag.err; exit 1; }
  python - $O/cloud_$tag.log $O/rt_$tag.json <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
print('cloud 1e8: %.2f ms  ' % d['ms_per_step'] + '  '.join(f"{s['stage'].split('(')[0]}={s['ms_per_step']:.2f}" for s in d['stages']))
r = json.loads([x for x in open(sys.argv[2]) if x.startswith('{')][-1])
print('streamed frame: p50 %.2f ms  p99 %.2f ms' % (r['p50_ms'], r['p99_ms']))
PY
fi
