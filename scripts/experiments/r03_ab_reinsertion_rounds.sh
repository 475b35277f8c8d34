#!/bin/bash
# parallel reinsertion rounds behind PLOC: nodes per ray, build time, throughput, parity
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_ai; mkdir -p $O; : > $O/log.txt
for r in 0 2 6 12; do echo -n "interior rounds $r: " >> $O/log.txt; GSP_BVH_TRACE=1 GSP_BVH_REINSERT=$r timeout 300 python scripts/experiments/reinsert_probe.py interior >> $O/log.txt 2>&1; done
for r in 0 6; do echo -n "caustics rounds $r: " >> $O/log.txt; GSP_BVH_REINSERT=$r timeout 300 python scripts/experiments/reinsert_probe.py caustics 2>&1 | tail -1 >> $O/log.txt; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 >> $O/log.txt
cat $O/log.txt
