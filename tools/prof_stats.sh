#!/bin/bash
# usage (GPU box): tools/prof_stats.sh <tag>   -- rocprofv3 kernel stats of a short bench run
tag=${1:-tmp}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cloud --no-stream --no-two-ranges > $R/gpurun_out/prof_$tag.log 2>&1 || { tail -5 $R/gpurun_out/prof_$tag.log; exit 1; }
echo ok
