#!/bin/bash
# usage (here, CPU): scripts/variant_sweep.sh build "8 3" "4 3" ... ; on GPU: scripts/variant_sweep.sh run
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then shift
  for cfg in "$@"; do set -- $cfg
    (cd pink_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -pragma-unroll-threshold=200000 \
       -DPINKHIP_GROUP=$1 -DPINKHIP_WAVES_PER_EU=$2 $3 pinkhip.hip -o libvariant_g$1w$2$4.so) &
  done; wait; ls pink_amd/csrc/libvariant_*
else
  for f in pink_amd/csrc/libvariant_*.so; do echo "== $f"; PINKHIP_LIBRARY=$f python scripts/gpu_time.py; done
fi
