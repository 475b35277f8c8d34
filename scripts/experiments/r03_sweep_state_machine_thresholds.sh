mkdir -p gpurun_out/r03_n; O=gpurun_out/r03_n
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "path_rays or living" 2>&1 | tail -3 > $O/log.txt
REPS=1 scripts/ab_quick.sh $O/ab.txt reps4 reps5 repl24 repl40 refill24 refill10 commit32 stall12
cat $O/log.txt $O/ab.txt
