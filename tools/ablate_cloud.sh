#!/bin/bash
# timing experiments on the 1e8-point cloud (wrong results): needs `make -C hope_amd/csrc ablate` beforehand
LIB=$GRAFT_REPO_ROOT/hope_amd/csrc/libhope_hip_ablate.so
[ -f "$LIB" ] || { echo "build it first: make -C hope_amd/csrc ablate"; exit 1; }
cd $GRAFT_REPO_ROOT
for a in "$@"; do
  HOPE_DEBUG=1 HOPE_DEBUG_ABLATE=$a timeout -k 10 200 python3 bench.py --workload cloud --no-cpu-baseline --hip-lib "$LIB" > gpurun_out/ablc_$a.log 2>&1 || { tail -5 gpurun_out/ablc_$a.log; exit 1; }
  python3 - "$a" <<'PY'
import json, sys
a = sys.argv[1]
d = json.loads([x for x in open(f'gpurun_out/ablc_{a}.log') if x.startswith('{')][-1])
print(f"ablate {a}: step {d['ms_per_step']:.2f} ms  " + '  '.join(f"{s['stage'].split('(')[0]}={s['ms_per_step']:.2f}" for s in d['stages'] if s['stage'].startswith(('labels', 'segments'))))
PY
done
