#!/bin/bash
# 8 waves per SIMD for k_trace on the final kernels (69 / 66 VGPRs at 7 waves: 5 / 3 registers spilled at 8), 12 LDS stack levels
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_ap; mkdir -p $O
bash scripts/ab_quick.sh $O/ab.txt lds12 tw8 tw8any
cat $O/ab.txt
