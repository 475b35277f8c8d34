mkdir -p gpurun_out/r03_g; O=gpurun_out/r03_g
V=$PWD/gpuspectral_amd/lib/variants
echo "== parity current (sum travels with the path)" > $O/log.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_textures.py -m gpu -x -q 2>&1 | tail -3 >> $O/log.txt
REPS=1 scripts/ab_quick.sh $O/ab.txt r02
cat $O/log.txt $O/ab.txt
