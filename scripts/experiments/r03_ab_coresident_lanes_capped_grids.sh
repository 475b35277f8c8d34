#!/bin/bash
# Co-resident kernels: GSP_LANES=2 with k_trace grids capped below the chip (5 / 4 / 6 blocks per CU instead of 7), so that the other lane's
# k_shade (33 KB LDS, 128 VGPRs per block-wave) or k_trace blocks fit beside them.
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_z; mkdir -p $O; : > $O/log.txt
for round in 1 2; do
for v in current bpc6 bpc5 bpc4; do
  for lanes in 1 2; do
    echo -n "$v lanes=$lanes: " >> $O/log.txt
    if [ $v = current ]; then GSP_LANES=$lanes timeout 300 python scripts/experiments/lanes_probe.py 2>&1 | tail -1 >> $O/log.txt
    else GSP_LANES=$lanes GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so timeout 300 python scripts/experiments/lanes_probe.py 2>&1 | tail -1 >> $O/log.txt; fi
  done
done
done
