#!/bin/bash
# both trace kernels at 8 waves per SIMD without spills (axis masks and the second triangle group packed into one register each)
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_as; mkdir -p $O
GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/both8.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not cli and not cpp_host" 2>&1 | tail -2 > $O/parity.txt
bash scripts/ab_quick.sh $O/ab.txt packst7 both8 both8t32
cat $O/parity.txt $O/ab.txt
