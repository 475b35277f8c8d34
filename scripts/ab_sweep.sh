#!/bin/bash
# On the GPU box: scripts/ab_probe.py for the current build and every variant named sw_* (two rounds), one line each.
# usage: scripts/ab_sweep.sh <outfile>
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/sweep.txt}; mkdir -p "$(dirname $OUT)"; : > $OUT
for round in 1 2; do
  echo -n "current: " >> $OUT; timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $OUT
  for f in gpuspectral_amd/lib/variants/sw_*.so; do
    v=$(basename $f .so); echo -n "$v: " >> $OUT
    GSP_LIB_PATH=$PWD/$f REPS=1 timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $OUT
  done
done
cat $OUT
