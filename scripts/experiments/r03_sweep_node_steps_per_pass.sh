#!/bin/bash
# node steps per bookkeeping pass, closest hit (c) / any hit (a), on the final kernels
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_ah; mkdir -p $O
bash scripts/ab_quick.sh $O/ab.txt c5a7 c5a10 c7a7 c7a10 c10a10 c7a10l16
cat $O/ab.txt
