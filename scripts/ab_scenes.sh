#!/bin/bash
# On the GPU box: per-scene kernel times of the current build and of the named variants (gpuspectral_amd/lib/variants/*.so), two
# rounds each, same box: scripts/ab_scenes.sh OUT "scene names" variant...
cd "$(dirname "$0")/.."
OUT=$1; SCENES=$2; shift; shift
: > $OUT
for round in 1 2; do
  echo "== current (round $round)" >> $OUT; SECONDS_=3 timeout 600 python tests/tools/scene_probe.py $SCENES 2>&1 | python scripts/probe_brief.py >> $OUT
  for v in "$@"; do
    echo "== $v (round $round)" >> $OUT
    GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so timeout 600 python tests/tools/scene_probe.py $SCENES 2>&1 | python scripts/probe_brief.py >> $OUT
  done
done
cat $OUT
