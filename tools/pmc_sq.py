"""Per-kernel SQ counter summary from rocprofv3 --pmc passes:
   python tools/pmc_sq.py gpurun_out/pmc_sq1 [gpurun_out/pmc_sq2 ...] [filter]
Prints, per kernel (average per launch): every collected counter, plus derived figures."""
import csv
import glob
import os
import sys
from collections import defaultdict

dirs = [a for a in sys.argv[1:] if os.path.isdir(a)]
flt = [a for a in sys.argv[1:] if not os.path.isdir(a)]
val = defaultdict(lambda: defaultdict(float))
calls = defaultdict(lambda: defaultdict(int))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "hope::" not in k and "rocprim" not in k:
                continue
            val[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k][r["Counter_Name"]] += 1
for k in sorted(val, key=lambda k: -val[k].get("SQ_BUSY_CYCLES", 0)):
    short = k.replace("void ", "").replace("hope::", "").split("(")[0][:70]
    if flt and not any(x in short for x in flt):
        continue
    c = {n: val[k][n] / max(calls[k][n], 1) for n in val[k]}
    print(short)
    print("   " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
    w = c.get("SQ_WAVES", 0)
    if w:
        parts = [f"per wave: valu {c.get('SQ_INSTS_VALU', 0) / w:.0f} salu {c.get('SQ_INSTS_SALU', 0) / w:.0f} "
                 f"lds {c.get('SQ_INSTS_LDS', 0) / w:.0f} vmem {c.get('SQ_INSTS_VMEM', 0) / w:.0f} "
                 f"cycles {c.get('SQ_WAVE_CYCLES', 0) / w:.0f} wait_inst {c.get('SQ_WAIT_INST_ANY', 0) / w:.0f}"]
        print("   " + parts[0])
    if c.get("SQ_ACTIVE_INST_VALU") and c.get("SQ_THREAD_CYCLES_VALU"):
        print(f"   lanes active per VALU instruction: {c['SQ_THREAD_CYCLES_VALU'] / c['SQ_ACTIVE_INST_VALU']:.1f} / 64")
