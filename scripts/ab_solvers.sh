#!/bin/bash
# A/B of the two stack+solve kernels on the library as built: sweep tableau (default) vs Goldfarb-Idnani (PINKHIP_SOLVER=packed)
# over the BASELINE shapes.  Usage (GPU box): bash scripts/ab_solvers.sh [lib]
LIB=${1:-pink_amd/csrc/libpinkhip.so}
for cfg in "draco3 tight 65536" "draco3 kinematic 65536" "jvrc tight 65536" "jvrc_noposture tight 65536" "draco3b tight 65536" "ur5 tight 4096" "ur5 tight 65536" "custom tight 65536"; do
  set -- $cfg
  for solver in sweep packed; do
    echo -n "$1 $2 B=$3 $solver: "
    AB_CONFIG=$1 AB_BOUNDS=$2 AB_BATCH=$3 PINKHIP_SOLVER=$solver python scripts/ab_variants.py $LIB | tail -1
  done
done
