#!/bin/bash
# leaf step: the triangle test under the lane's condition again (temporaries only leave the block), updates as selects outside
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_at; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 > $O/parity.txt
bash scripts/ab_quick.sh $O/ab.txt prev
bash scripts/pmc_quick.sh cur gpuspectral_amd/lib/libgpuspectral_pt.so 2>&1 | grep " a " | cut -c1-330 > $O/pmc.txt
bash scripts/pmc_quick.sh prev gpuspectral_amd/lib/variants/prev.so 2>&1 | grep " a " | cut -c1-330 >> $O/pmc.txt
cat $O/parity.txt $O/ab.txt $O/pmc.txt
