mkdir -p gpurun_out/r03_k; O=gpurun_out/r03_k
V=$PWD/gpuspectral_amd/lib/variants
echo "== parity current (triangle-group ring, 4 entries)" > $O/log.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_textures.py -m gpu -x -q 2>&1 | tail -3 >> $O/log.txt
for v in current pre_ring tq8_lbc48; do
  echo -n "stats $v: " >> $O/log.txt
  if [ $v = current ]; then timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt; else GSP_LIB_PATH=$V/$v.so timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt; fi
done
REPS=1 scripts/ab_quick.sh $O/ab.txt pre_ring tq2 tq8 tq4_lbc40 tq4_lbc44 tq8_lbc48
cat $O/log.txt $O/ab.txt
