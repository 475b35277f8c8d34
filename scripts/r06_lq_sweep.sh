set -u
O=gpurun_out/r06g; mkdir -p $O
V=$PWD/gpuspectral_amd/lib/variants
OUT=$O/$1.txt; : > $OUT
for round in 1 2; do
  echo -n "current: " >> $OUT; timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $OUT
  for f in $V/*.so; do v=$(basename $f .so)
    echo -n "$v: " >> $OUT; GSP_LIB_PATH=$f timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $OUT
  done
done
cat $OUT
