mkdir -p gpurun_out/r03_p; O=gpurun_out/r03_p
V=$PWD/gpuspectral_amd/lib/variants
: > $O/log.txt
for v in shboth; do echo "== parity $v" >> $O/log.txt; GSP_LIB_PATH=$V/$v.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not cli and not cpp_host" 2>&1 | tail -2 >> $O/log.txt; done
REPS=1 scripts/ab_quick.sh $O/ab.txt shgather shbar shboth shgather5
cat $O/log.txt $O/ab.txt
