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
  HOPE_SUB_BATCHES=$sb timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_${tag}_sb$sb.log 2>&1 || { tail -5 $O/bench_${tag}_sb$sb.log; exit 1; }
  python - $O/bench_${tag}_sb$sb.log <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith('{')][-1])
print(sys.argv[1].split('/')[-1], 'value %.4g pts/s  %.3f ms/step' % (d['value'], d['ms_per_step']))
print('  ' + '  '.join(f"{s['stage'].split('(')[0]}={s['ms_per_step']:.2f}" for s in d['stages']))
PY
done
