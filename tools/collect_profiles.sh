#!/bin/bash
# usage (here, after `gpurun -- tools/profile_round.sh <tag>`): tools/collect_profiles.sh <tag>
# copies the summaries of gpurun_out/ (scratch) into profiles/ (tracked) and regenerates the traffic tables
tag=${1:?tag}
cd "$(dirname "$0")/.."
O=gpurun_out
cp $O/bench_$tag.json profiles/${tag}_bench.json
cp $O/prof_$tag/*/*kernel_stats.csv profiles/${tag}_bench_kernel_stats.csv
cp $O/prof_cloud_$tag/*/*kernel_stats.csv profiles/${tag}_cloud100M_kernel_stats.csv
cp $O/cloud5M_$tag.json profiles/${tag}_cloud_loft5M.json
python3 tools/pmc_traffic.py $O/pmc_fetch_$tag $O/pmc_write_$tag 3 hbm_traffic.json > profiles/${tag}_hbm_traffic.md
python3 tools/pmc_traffic.py $O/pmc_fetch_cloud_$tag $O/pmc_write_cloud_$tag 3 hbm_traffic_cloud.json > profiles/${tag}_hbm_traffic_cloud.md
ls -la profiles/${tag}_*
