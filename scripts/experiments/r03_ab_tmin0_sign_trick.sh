mkdir -p gpurun_out/r03_r; O=gpurun_out/r03_r
echo "== parity (tmin 0 sign trick)" > $O/log.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py -m gpu -x -q 2>&1 | tail -3 >> $O/log.txt
REPS=1 scripts/ab_quick.sh $O/ab.txt prev
cat $O/log.txt $O/ab.txt
