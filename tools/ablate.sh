#!/bin/bash
# timing experiments (wrong results): HOPE_DEBUG_ABLATE = 10*knn_mode + normals_mode
# needs the ablation build of the library, made beforehand on the CPU box: make -C hope_amd/csrc ablate
LIB=$(cd "$(dirname "$0")/.." && pwd)/hope_amd/csrc/libhope_hip_ablate.so
[ -f "$LIB" ] || { echo "build it first: make -C hope_amd/csrc ablate"; exit 1; }
for a in "$@"; do
  HOPE_DEBUG=1 HOPE_DEBUG_ABLATE=$a timeout -k 10 100 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cloud --no-stream --no-two-ranges --hip-lib "$LIB" > gpurun_out/abl_$a.log 2>&1 || exit 1
  python3 - "$a" <<'PY'
import json, sys
a = sys.argv[1]
d = json.loads([x for x in open(f'gpurun_out/abl_{a}.log') if x.startswith('{')][-1])
s = {x['stage']: x['ms_per_step'] for x in d['stages']}
print(f"ablate {a}: step {d['ms_per_step']:.2f} ms  normals {s.get('k_normals')}  knn {s.get('k_knn')} todo {s.get('k_knn_todo')}")
PY
done
