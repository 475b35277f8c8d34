#!/bin/bash
# leaf step without divergent blocks + rotation-mask triangle test: parity, then A/B against HEAD's build on the same box
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_aa; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5 > $O/parity.txt
bash scripts/ab_quick.sh $O/ab.txt base top0 top32l18
cat $O/parity.txt $O/ab.txt
