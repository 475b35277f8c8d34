#!/bin/bash
# any-hit kernel at 8 waves per SIMD without spills (the three axis masks of the triangle test packed into one register between leaf steps)
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_ar; mkdir -p $O
GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/any8p.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not cli and not cpp_host" 2>&1 | tail -2 > $O/parity.txt
bash scripts/ab_quick.sh $O/ab.txt pack7 any8p any8p14
cat $O/parity.txt $O/ab.txt
