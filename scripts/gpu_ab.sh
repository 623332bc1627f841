#!/bin/bash
# A/B of kernel variants on the GPU box: scripts/gpu_ab.sh lib1.so lib2.so ...  (headline regime, then tracking)
export TMPDIR=/tmp
mkdir -p gpurun_out
(for r in tight tracking; do echo "== $r"; AB_BOUNDS=$r python scripts/ab_variants.py "$@"; done) > gpurun_out/ab_variants.txt 2>&1
cat gpurun_out/ab_variants.txt
