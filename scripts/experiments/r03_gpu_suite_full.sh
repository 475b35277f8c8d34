mkdir -p gpurun_out/r03_i; O=gpurun_out/r03_i
timeout 2400 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -15 > $O/log.txt
cat $O/log.txt
