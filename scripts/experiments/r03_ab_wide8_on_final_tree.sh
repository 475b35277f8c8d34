#!/bin/bash
# the 8-wide variant rebuilt from the final tree (reinsertion rounds, straight-line leaf step; with / without the LDS node copy)
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_am; mkdir -p $O; : > $O/log.txt
for v in w8n w8ntop; do echo -n "stats $v: " >> $O/log.txt; GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt; done
bash scripts/ab_quick.sh $O/ab.txt w8n w8ntop
cat $O/ab.txt >> $O/log.txt
GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/w8ntop.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not cli and not cpp_host" 2>&1 | tail -3 >> $O/log.txt
cat $O/log.txt
