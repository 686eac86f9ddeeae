"""Prints the hope:: and rocprim rows of a rocprofv3 kernel_stats.csv (per-call average in ms)."""
import csv
import glob
import sys

path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True) if not path.endswith(".csv") else [path]
calls_per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for f in files:
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "hope::" in n or "rocprim" in n:
            short = n.replace("void ", "").replace("hope::", "")
            short = short.split("(")[0][:70] if "rocprim" not in n else "rocprim::" + n.split("detail::")[-1][:60]
            print(f"{short:72s} calls={int(r['Calls']):5d} avg_ms={float(r['AverageNs'])/1e6:9.4f} "
                  f"total_ms={float(r['TotalDurationNs'])/1e6:9.3f}")
