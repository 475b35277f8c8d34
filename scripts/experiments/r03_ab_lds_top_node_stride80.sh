#!/bin/bash
# LDS copy of the top node records with an odd record stride (80 B: all 16 bank groups) against 64 B
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_ae; mkdir -p $O
bash scripts/ab_quick.sh $O/ab.txt top0 top56 pad56
cat $O/ab.txt
