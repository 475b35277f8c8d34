# r06: cost side of a wave-level leaf queue (machinery proxy at 6 waves per SIMD) + read-back timing + the edit tests on the new library
set -u
O=gpurun_out/r06c; mkdir -p $O
V=$PWD/gpuspectral_amd/lib/variants
: > $O/ab_lq_proxy.txt
for round in 1 2 3; do
  echo -n "current: " >> $O/ab_lq_proxy.txt; timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/ab_lq_proxy.txt
  for v in base6 lq6p2 lq6p3; do
    echo -n "$v: " >> $O/ab_lq_proxy.txt; GSP_LIB_PATH=$V/$v.so timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/ab_lq_proxy.txt
  done
done
cat $O/ab_lq_proxy.txt
timeout 300 python tests/tools/download_probe.py > $O/download_probe.txt 2>&1; cat $O/download_probe.txt
timeout 1500 python -m pytest tests/test_gpu_scene_updates.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_updates.txt
