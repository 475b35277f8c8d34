#!/bin/bash
# thresholds of the wave state machine re-swept on the final kernels (cheaper leaf step, LDS top nodes), and the pool size
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_ag; mkdir -p $O
bash scripts/ab_quick.sh $O/ab.txt lb24 lb40 sb4 sb16 reps4 reps7 lba14 lba28 rf8 rf24
( for p in 67108864 100663296 134217728 201326592; do echo -n "pool $p: "; GSP_POOL_PATHS=$p SPP=256 REPS=2 timeout 300 python scripts/experiments/lanes_probe.py 2>&1 | tail -1; done ) > $O/pool.txt 2>&1
cat $O/ab.txt $O/pool.txt
