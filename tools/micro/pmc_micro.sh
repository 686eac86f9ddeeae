#!/bin/bash
# usage (GPU box): tools/micro/pmc_micro.sh <tag> <kernel name fragment> <program and arguments...>
# SQ / TCP counters of the kernels of a micro-benchmark whose name holds the fragment, one rocprofv3 --pmc pass per group.
tag=$1; frag=$2; shift 2
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/pmcm_${tag}_$i -- "$@" > $O/pmcm_${tag}_$i.log 2>&1 || { tail -5 $O/pmcm_${tag}_$i.log; exit 1; }
done
python3 - $O $tag "$frag" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
O, tag, frag = sys.argv[1:4]
tot = defaultdict(lambda: defaultdict(float)); calls = defaultdict(lambda: defaultdict(int))
for d in sorted(glob.glob(os.path.join(O, f"pmcm_{tag}_*"))):
    if not os.path.isdir(d): continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if frag in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0][-60:]
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
with open(os.path.join(O, f"pmcm_{tag}.txt"), "w") as out:
    for k in sorted(tot):
        c = {n: tot[k][n] / max(1, calls[k][n]) for n in tot[k]}
        w = max(c.get("SQ_WAVES", 1), 1)
        lines = [k, "  " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())),
                 f"  per wave: valu {c.get('SQ_INSTS_VALU',0)/w:.0f} salu {c.get('SQ_INSTS_SALU',0)/w:.0f} lds {c.get('SQ_INSTS_LDS',0)/w:.0f} "
                 f"vmem_rd {c.get('SQ_INSTS_VMEM_RD',0)/w:.0f} smem {c.get('SQ_INSTS_SMEM',0)/w:.0f} wave_cycles {c.get('SQ_WAVE_CYCLES',0)/w:.0f}; "
                 f"lanes/VALU {c.get('SQ_THREAD_CYCLES_VALU',0)/max(c.get('SQ_ACTIVE_INST_VALU',1),1):.1f}; "
                 f"TCP accesses/wave {c.get('TCP_TOTAL_CACHE_ACCESSES_sum',0)/w:.0f}"]
        for l in lines:
            print(l); out.write(l + "\n")
PY
