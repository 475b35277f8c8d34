mkdir -p gpurun_out/r03_l; O=gpurun_out/r03_l
V=$PWD/gpuspectral_amd/lib/variants
( echo "== current"; python scripts/experiments/living_debug2.py 2>&1 | tail -40
  echo "== pre_ring"; GSP_LIB_PATH=$V/pre_ring.so python scripts/experiments/living_debug2.py 2>&1 | tail -40
  echo "== r02"; GSP_LIB_PATH=$V/r02.so python scripts/experiments/living_debug2.py 2>&1 | tail -40 ) > $O/log.txt 2>&1
cat $O/log.txt
