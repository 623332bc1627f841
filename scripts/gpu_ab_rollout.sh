#!/bin/bash
# A/B of the whole-step kernel between libraries: scripts/gpu_ab_rollout.sh lib1.so lib2.so ...  (nv = 30 floating base)
export TMPDIR=/tmp
mkdir -p gpurun_out
for lib in "$@"; do
  echo "== $lib"
  PINKHIP_LIBRARY=$PWD/$lib ROLLOUT_ONLY=${ROLLOUT_ONLY:-floating_base_nv30} python scripts/rollout_bench.py 2>&1 | python -c "
import sys, json
txt = sys.stdin.read()
try:
    d = json.loads(txt[txt.index('{'):])
    for k, v in d.items():
        print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if not isinstance(vv, (list, dict))})
except Exception as e:
    print(txt[-2000:])
"
done | tee gpurun_out/ab_rollout.txt
