#!/bin/bash
# with the reinsertion rounds: greedy vs parity collapse; rounds 4 / 6 / 8
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_aj; mkdir -p $O; : > $O/log.txt
for c in parity greedy; do for r in 4 6 8; do echo -n "interior collapse $c rounds $r: " >> $O/log.txt; GSP_COLLAPSE=$c GSP_BVH_REINSERT=$r timeout 300 python scripts/experiments/reinsert_probe.py interior 2>&1 | tail -1 >> $O/log.txt; done; done
echo -n "caustics collapse greedy rounds 6: " >> $O/log.txt; GSP_COLLAPSE=greedy GSP_BVH_REINSERT=6 timeout 300 python scripts/experiments/reinsert_probe.py caustics 2>&1 | tail -1 >> $O/log.txt
cat $O/log.txt
