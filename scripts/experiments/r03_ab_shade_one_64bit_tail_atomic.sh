#!/bin/bash
# k_shade: one 64-bit atomic per tile for both queue tails instead of two 32-bit ones on the same line
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_ao; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 > $O/parity.txt
bash scripts/ab_quick.sh $O/ab.txt prev
cat $O/parity.txt $O/ab.txt
