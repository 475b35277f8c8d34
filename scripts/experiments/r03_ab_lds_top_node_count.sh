#!/bin/bash
# how many records of the top of the tree to keep in LDS (against stack levels, 7 blocks x <= 22.8 KB per CU)
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_ad; mkdir -p $O
python scripts/stats_probe.py 2>&1 | tail -1 > $O/depth.txt
bash scripts/ab_quick.sh $O/ab.txt base top0 top24l19 top88l15 top128l13
cat $O/ab.txt
