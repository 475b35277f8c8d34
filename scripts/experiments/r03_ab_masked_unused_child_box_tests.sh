#!/bin/bash
# node step: box tests of child positions 3 and 2 only in the lanes whose node uses them (exec-masked), same verdicts
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_au; mkdir -p $O
GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/maskch.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "not cli and not cpp_host" 2>&1 | tail -2 > $O/parity.txt
bash scripts/ab_quick.sh $O/ab.txt maskch
bash scripts/pmc_quick.sh cur gpuspectral_amd/lib/libgpuspectral_pt.so 2>&1 | grep " a " | cut -c1-330 > $O/pmc.txt
bash scripts/pmc_quick.sh maskch gpuspectral_amd/lib/variants/maskch.so 2>&1 | grep " a " | cut -c1-330 >> $O/pmc.txt
cat $O/parity.txt $O/ab.txt $O/pmc.txt
