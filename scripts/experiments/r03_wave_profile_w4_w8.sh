mkdir -p gpurun_out/r03_c; O=gpurun_out/r03_c
V=$PWD/gpuspectral_amd/lib/variants
echo "== wave profile w4prof" > $O/log.txt
GSP_LIB_PATH=$V/w4prof.so timeout 300 python scripts/wave_profile.py >> $O/log.txt 2>&1
for v in post0any post0both; do
  echo -n "stats $v: " >> $O/log.txt
  GSP_LIB_PATH=$V/$v.so timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt
done
echo "== parity w8_6" >> $O/log.txt
GSP_LIB_PATH=$V/w8_6.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -m gpu -x -q -k "not cli and not cpp_host" 2>&1 | tail -3 >> $O/log.txt
REPS=1 scripts/ab_quick.sh $O/ab.txt r02 post0any post0both lb12 lbc24 lbc40 w8_6 w8_5
cat $O/log.txt $O/ab.txt
