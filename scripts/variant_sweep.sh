#!/bin/bash
# usage (here, CPU): scripts/variant_sweep.sh build "name -DFLAG=.. -DFLAG2=.." ... ; on GPU: scripts/variant_sweep.sh run [ncases]
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then shift
  for cfg in "$@"; do set -- $cfg; name=$1; shift
    (cd pink_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -pragma-unroll-threshold=200000 \
       "$@" pinkhip.hip -o libvariant_$name.so) &
  done; wait; ls pink_amd/csrc/libvariant_*
else
  for f in pink_amd/csrc/libvariant_*.so; do echo "== $f"; PINKHIP_LIBRARY=$f python scripts/gpu_time.py $2; done
fi
