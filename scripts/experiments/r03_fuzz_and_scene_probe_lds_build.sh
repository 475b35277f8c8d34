#!/bin/bash
# final build of the round (LDS copy of the top nodes, straight-line leaf step): fresh fuzz seeds + the per-scene probe
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_af; mkdir -p $O
( timeout 1500 python tests/tools/fuzz_parity.py 3000 700000 2>&1 | tail -1
  timeout 900 python tests/tools/fuzz_parity.py 1000 800000 dormant 2>&1 | tail -1
  GSP_FINISH_PATHS=0 timeout 600 python tests/tools/fuzz_parity.py 500 900000 2>&1 | tail -1 ) > $O/fuzz.txt 2>&1
timeout 900 python tests/tools/scene_probe.py coffee staircase2 living-room interior caustics materials cornell-box > $O/scenes.txt 2>&1
cat $O/fuzz.txt; cut -c1-330 $O/scenes.txt
