# r06: three knobs of the rejected leaf-queue kernel (variants built from commit 62b6528), for the record
set -u
O=gpurun_out/r06g; mkdir -p $O
V=$PWD/gpuspectral_amd/lib/variants
: > $O/ab_lq_knobs.txt
for round in 1 2; do
  echo -n "current: " >> $O/ab_lq_knobs.txt; timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/ab_lq_knobs.txt
  for v in lq3 lqn32 lqn48 lqn32c8; do
    echo -n "$v: " >> $O/ab_lq_knobs.txt; GSP_LIB_PATH=$V/$v.so timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/ab_lq_knobs.txt
  done
done
cat $O/ab_lq_knobs.txt
