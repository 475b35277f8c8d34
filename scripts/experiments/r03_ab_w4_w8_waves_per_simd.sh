mkdir -p gpurun_out/r03_d; O=gpurun_out/r03_d
V=$PWD/gpuspectral_amd/lib/variants
: > $O/log.txt
echo -n "stats r02 greedy: " >> $O/log.txt; GSP_COLLAPSE=greedy GSP_LIB_PATH=$V/r02.so timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt
echo -n "stats current parity: " >> $O/log.txt; GSP_COLLAPSE=parity timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt
echo -n "stats current greedy: " >> $O/log.txt; timeout 300 python scripts/stats_probe.py 2>&1 | tail -1 >> $O/log.txt
for i in 1 2; do
echo -n "r02 greedy: " >> $O/log.txt; GSP_COLLAPSE=greedy GSP_LIB_PATH=$V/r02.so timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/log.txt
echo -n "r02: " >> $O/log.txt; GSP_LIB_PATH=$V/r02.so timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/log.txt
echo -n "current parity: " >> $O/log.txt; GSP_COLLAPSE=parity timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/log.txt
echo -n "current: " >> $O/log.txt; timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/log.txt
done
cat $O/log.txt
