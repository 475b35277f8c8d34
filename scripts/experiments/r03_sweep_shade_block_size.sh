#!/bin/bash
# k_shade block size (tile of the BSDF-type sort, barrier scope, one atomic pair per tile): 64 / 128 / 256 (default) / 512 threads
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_an; mkdir -p $O
bash scripts/ab_quick.sh $O/ab.txt shb64 shb128 shb512
cat $O/ab.txt
