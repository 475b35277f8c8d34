mkdir -p gpurun_out/r03_u; O=gpurun_out/r03_u
: > $O/log2.txt
for round in 1 2; do
for p in 50331648 100663296 134217728 201326592; do
  echo -n "pool $p: " >> $O/log2.txt; GSP_POOL_PATHS=$p timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/log2.txt
done
done
echo -n "pool 134217728 ring 32G: " >> $O/log2.txt; GSP_RING_BYTES=34359738368 GSP_POOL_PATHS=134217728 timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/log2.txt
cat $O/log2.txt
