# r06: the wave-level leaf queue experiment (scripts/experiments/r06_pt_wavetrace_lq.h): parity, then A/B on one box
set -u
O=gpurun_out/r06d; mkdir -p $O
V=$PWD/gpuspectral_amd/lib/variants
for v in lq3; do
  echo "== parity $v" | tee -a $O/parity.txt
  GSP_LIB_PATH=$V/$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not cli and not cpp_host" 2>&1 | tail -15 | tee -a $O/parity.txt
done
: > $O/ab_lq.txt
for round in 1 2 3; do
  echo -n "current: " >> $O/ab_lq.txt; timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/ab_lq.txt
  for v in lq1 lq2 lq3; do
    echo -n "$v: " >> $O/ab_lq.txt; GSP_LIB_PATH=$V/$v.so timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/ab_lq.txt
  done
done
cat $O/ab_lq.txt
