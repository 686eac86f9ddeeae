"""Prints VGPR / SGPR / scratch / LDS / occupancy per kernel of hope_hip.hip (hipcc -Rpass-analysis)."""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "hope_amd", "csrc", "hope_hip.hip")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
pat = sys.argv[1] if len(sys.argv) > 1 else ""
cur = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(?:Function )?Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        continue
    m = re.search(r"remark:\s+(\w[\w ]*\w)\s*(?:\[bytes/\w+\])?: (\d+)", line)
    if m and cur:
        cur[m.group(1).strip()] = int(m.group(2))
        if m.group(1).startswith("LDS Size"):
            n = cur["name"].split("(")[0].replace("void hope::", "")
            if pat in n:
                print(f"{n[:60]:60s} vgpr={cur.get('VGPRs', -1):4d} agpr={cur.get('AGPRs', 0):3d} sgpr={cur.get('TotalSGPRs', -1):4d} "
                      f"scratch={cur.get('ScratchSize', 0):5d} occ={cur.get('Occupancy', -1):2d} lds={cur.get('LDS Size', 0)}")
            cur = {}
