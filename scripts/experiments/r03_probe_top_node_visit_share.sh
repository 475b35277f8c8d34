#!/bin/bash
# share of the node visits that fall into the first N records of the node array (the top levels, emitted first)
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_ac; mkdir -p $O; : > $O/log.txt
echo -n "all: " >> $O/log.txt; python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt
for n in 24 64 88 344; do echo -n "first $n: " >> $O/log.txt; GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/cnt$n.so python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt; done
cat $O/log.txt
