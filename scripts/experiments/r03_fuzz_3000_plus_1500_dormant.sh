mkdir -p gpurun_out/r03_j; O=gpurun_out/r03_j
( timeout 1500 python tests/tools/fuzz_parity.py 3000 200000 2>&1 | tail -3
  timeout 900 python tests/tools/fuzz_parity.py 1500 300000 dormant 2>&1 | tail -3
  GSP_FINISH_PATHS=0 timeout 600 python tests/tools/fuzz_parity.py 800 400000 2>&1 | tail -3
  GSP_PRIMARY_MEMO=0 timeout 600 python tests/tools/fuzz_parity.py 800 500000 2>&1 | tail -3
  GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/w8_6.so timeout 600 python tests/tools/fuzz_parity.py 800 600000 2>&1 | tail -3 ) > $O/fuzz.txt 2>&1
timeout 900 python tests/tools/scene_probe.py cornell-box coffee staircase2 living-room interior materials caustics > $O/scene_probe.txt 2>&1
cat $O/fuzz.txt $O/scene_probe.txt
