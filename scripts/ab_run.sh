#!/bin/bash
# On the GPU box: parity subset + throughput probe for each build variant in gpuspectral_amd/lib/variants and the current build.
# usage: scripts/ab_run.sh <outfile> [variant names...]
cd "$(dirname "$0")/.."
OUT=$1; shift
names=("$@")
if [ ${#names[@]} -eq 0 ]; then for f in gpuspectral_amd/lib/variants/*.so; do names+=("$(basename $f .so)"); done; fi
: > $OUT
for v in "${names[@]}"; do
  echo "== parity $v" >> $OUT
  GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -m gpu -x -q -k "not cli and not cpp_host" 2>&1 | tail -2 >> $OUT
done
for round in 1 2 3; do
  echo -n "current: " >> $OUT; timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $OUT
  for v in "${names[@]}"; do
    echo -n "$v: " >> $OUT
    GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $OUT
  done
done
cat $OUT
