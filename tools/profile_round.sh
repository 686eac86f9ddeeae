#!/bin/bash
# usage (GPU box): tools/profile_round.sh <tag>  -- everything profiles/ holds for one build:
# the default bench line (frames headline + cloud + stream sub-objects, CPU baselines), rocprofv3 kernel stats of the
# frames step and of the 1e8-point cloud step, the FETCH_SIZE / WRITE_SIZE passes of both (each in its own run), and
# the 5M-point loft stand-in.
tag=${1:-tmp}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
timeout -k 10 400 python $R/bench.py > $O/bench_$tag.json 2> $O/bench_$tag.err || { tail -5 $O/bench_$tag.err; exit 1; }
echo "bench done"
cd /tmp && export TMPDIR=/tmp
F="--no-cpu-baseline --no-cloud --no-stream --no-two-ranges"
C="--workload cloud --no-cpu-baseline"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $R/bench.py --steps 3 --warmup 1 $F > $O/prof_$tag.log 2>&1 || { tail -5 $O/prof_$tag.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cloud_$tag -- python3 $R/bench.py --steps 3 --warmup 1 $C > $O/prof_cloud_$tag.log 2>&1 || { tail -5 $O/prof_cloud_$tag.log; exit 1; }
echo "stats done"
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$tag -- python3 $R/bench.py --steps 2 --warmup 1 $F > $O/pmc_fetch_$tag.log 2>&1 || { tail -5 $O/pmc_fetch_$tag.log; exit 1; }
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_$tag -- python3 $R/bench.py --steps 2 --warmup 1 $F > $O/pmc_write_$tag.log 2>&1 || { tail -5 $O/pmc_write_$tag.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_cloud_$tag -- python3 $R/bench.py --steps 2 --warmup 1 $C > $O/pmc_fetch_cloud_$tag.log 2>&1 || { tail -5 $O/pmc_fetch_cloud_$tag.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_cloud_$tag -- python3 $R/bench.py --steps 2 --warmup 1 $C > $O/pmc_write_cloud_$tag.log 2>&1 || { tail -5 $O/pmc_write_cloud_$tag.log; exit 1; }
echo "pmc done"
cd $R
timeout -k 10 200 python tools/bench_cloud.py --kind loft --points 5000000 > $O/cloud5M_$tag.json 2> $O/cloud5M_$tag.err || { tail -5 $O/cloud5M_$tag.err; exit 1; }
echo "all done"
