mkdir -p gpurun_out/r03_b; O=gpurun_out/r03_b
echo "== parity current (W4T)" > $O/log.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -m gpu -x -q 2>&1 | tail -5 >> $O/log.txt
echo "== parity w8" >> $O/log.txt
GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/w8.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -m gpu -x -q -k "not cli and not cpp_host" 2>&1 | tail -5 >> $O/log.txt
for v in current r02 w8; do
  echo -n "stats $v: " >> $O/log.txt
  if [ $v = current ]; then timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt; else GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt; fi
done
scripts/ab_quick.sh $O/ab.txt r02 w8
cat $O/log.txt $O/ab.txt
