#!/bin/bash
# usage (GPU box): tools/pmc_kernel.sh <tag> <kernel name fragment>  -- SQ / TCP / TCC counters of one kernel of the
# frames bench, one rocprofv3 --pmc pass per counter group (each in its own run), summed over the kernel's launches.
tag=${1:-k}
frag=${2:-k_centroids}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE TCC_EA_RDREQ_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmck_${tag}_$i -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-cloud --no-stream --no-two-ranges > $O/pmck_${tag}_$i.log 2>&1 || { tail -5 $O/pmck_${tag}_$i.log; exit 1; }
  echo "pass $i done"
done
python3 - $O $tag "$frag" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
O, tag, frag = sys.argv[1:4]
tot = defaultdict(float); calls = defaultdict(int)
for d in sorted(glob.glob(os.path.join(O, f"pmck_{tag}_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if frag in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); calls[r["Counter_Name"]] += 1
with open(os.path.join(O, f"pmck_{tag}.txt"), "w") as out:
    for k in sorted(tot):
        line = f"{k:32s} {tot[k] / max(1, calls[k]):.4g} per launch ({calls[k]} launches)"
        print(line); out.write(line + "\n")
PY
