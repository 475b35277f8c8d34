mkdir -p gpurun_out/r03_m; O=gpurun_out/r03_m
timeout 2400 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -8 > $O/log.txt
for i in 1 2 3; do GSP_FINISH_PATHS=0 python scripts/experiments/living_debug.py 2>&1 | tail -3 >> $O/log.txt; done
python scripts/experiments/living_debug2.py 2>&1 | tail -9 >> $O/log.txt
for i in 1 2; do echo -n "current: " >> $O/log.txt; timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/log.txt; done
cat $O/log.txt
