mkdir -p gpurun_out/r03_f; O=gpurun_out/r03_f
V=$PWD/gpuspectral_amd/lib/variants
echo "== parity current (W4T, any-hit mask path, parity collapse)" > $O/log.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -m gpu -x -q 2>&1 | tail -3 >> $O/log.txt
echo -n "stats current: " >> $O/log.txt; timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt
REPS=1 scripts/ab_quick.sh $O/ab.txt r02 abl1 abl3 w8_6
cat $O/log.txt $O/ab.txt
