#!/bin/bash
# A/B of build variants on one box: scripts/ab_variants.sh [variant.so ...]   (default: all of lib/variants + current)
cd "$(dirname "$0")/.."
libs=("$@")
if [ ${#libs[@]} -eq 0 ]; then libs=(gpuspectral_amd/lib/variants/*.so gpuspectral_amd/lib/libgpuspectral_pt.so); fi
for round in 1 2; do
  for l in "${libs[@]}"; do
    echo -n "$(basename $l): "
    GSP_LIB_PATH=$PWD/$l timeout 200 python scripts/ab_probe.py 2>&1 | tail -1
  done
done
