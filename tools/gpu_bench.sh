#!/bin/bash
# usage: tools/gpu_bench.sh [tag]   -- run GPU parity tests + bench (+ rocprof kernel stats) on the MI355X box
tag=${1:-tmp}
make -C /root/repo/hope_amd/csrc 2>&1 | grep -E "error" && exit 1
/usr/local/graft/bin/gpurun --timeout 900 -- "timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log; timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cloud --no-stream --no-two-ranges > gpurun_out/bench_$tag.log 2>&1 && cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d \$GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 \$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cloud --no-stream --no-two-ranges > \$GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1" 2>&1 | tail -6
python - <<PY
import json
l=[x for x in open('/root/repo/gpurun_out/bench_$tag.log') if x.startswith('{')]
if not l:
    print(open('/root/repo/gpurun_out/bench_$tag.log').read()[-3000:])
else:
    d=json.loads(l[-1])
    print('value %.4g pts/s  %.3f ms/step' % (d['value'], d['ms_per_step']))
    for s in d['stages']: print(f"{s['stage']:28s} {s['ms_per_step']:8.3f} ms  {s['achieved_GBs']:8.1f} GB/s")
PY
python /root/repo/tools/prof_summary.py /root/repo/gpurun_out/prof_$tag 3 | head -${2:-14}
