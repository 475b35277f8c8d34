mkdir -p gpurun_out/r03_q; O=gpurun_out/r03_q
echo "== parity (slim connect commit)" > $O/log.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_textures.py -m gpu -x -q 2>&1 | tail -3 >> $O/log.txt
REPS=1 scripts/ab_quick.sh $O/ab.txt prev nounroll
cat $O/log.txt $O/ab.txt
