mkdir -p gpurun_out/r03_s; O=gpurun_out/r03_s
V=$PWD/gpuspectral_amd/lib/variants
echo "== parity any8" > $O/log.txt
GSP_LIB_PATH=$V/any8.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not cli and not cpp_host" 2>&1 | tail -2 >> $O/log.txt
REPS=1 scripts/ab_quick.sh $O/ab.txt any8
cat $O/log.txt $O/ab.txt
