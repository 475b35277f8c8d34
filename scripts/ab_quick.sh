#!/bin/bash
# On the GPU box: throughput probe (no parity run) of the current build and the named variants, 3 rounds.
cd "$(dirname "$0")/.."
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
: > $OUT
for round in 1 2 3; do
  echo -n "current: " >> $OUT; timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $OUT
  for v in "$@"; do
    echo -n "$v: " >> $OUT
    GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $OUT
  done
done
