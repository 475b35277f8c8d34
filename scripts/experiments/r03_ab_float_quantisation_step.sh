#!/bin/bash
# 4-wide nodes with any float as the quantisation step (extent / 255) instead of a power of two: parity, nodes per ray, A/B
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_aq; mkdir -p $O; : > $O/log.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -x -q -m gpu 2>&1 | tail -3 >> $O/log.txt
echo -n "stats current: " >> $O/log.txt; timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt
echo -n "stats prev: " >> $O/log.txt; GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/prev.so timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt
bash scripts/ab_quick.sh $O/ab.txt prev
cat $O/ab.txt >> $O/log.txt
SECONDS=2 timeout 900 python tests/tools/scene_probe.py coffee staircase2 living-room caustics 2>&1 | cut -c1-170 >> $O/log.txt
cat $O/log.txt
