#!/bin/bash
# wider fuzz on the final library of the round
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_av; mkdir -p $O
( timeout 2400 python tests/tools/fuzz_parity.py 12000 1000000 2>&1 | tail -1
  timeout 1200 python tests/tools/fuzz_parity.py 3000 1100000 dormant 2>&1 | tail -1
  GSP_FINISH_PATHS=0 timeout 900 python tests/tools/fuzz_parity.py 1500 1200000 2>&1 | tail -1
  GSP_PRIMARY_MEMO=0 timeout 900 python tests/tools/fuzz_parity.py 1500 1300000 2>&1 | tail -1
  GSP_BVH_REINSERT=0 timeout 900 python tests/tools/fuzz_parity.py 1500 1400000 2>&1 | tail -1
  GSP_BVH_REINSERT=20 timeout 900 python tests/tools/fuzz_parity.py 1500 1500000 2>&1 | tail -1
  GSP_LANES=2 timeout 900 python tests/tools/fuzz_parity.py 1000 1600000 2>&1 | tail -1 ) > $O/fuzz.txt 2>&1
cat $O/fuzz.txt
