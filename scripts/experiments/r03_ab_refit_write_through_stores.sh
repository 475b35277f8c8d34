#!/bin/bash
# refit of the reinsertion rounds through write-through stores instead of __threadfence(): build time, same tree, parity
cd "$(dirname "$0")/../.."
O=gpurun_out/r03_al; mkdir -p $O; : > $O/log.txt
for r in 6 0; do echo -n "rounds $r: " >> $O/log.txt; GSP_BVH_TRACE=1 GSP_BVH_REINSERT=$r timeout 300 python scripts/experiments/reinsert_probe.py interior 2>&1 | tail -1 >> $O/log.txt; done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3 >> $O/log.txt
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/scripts/stats_probe.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - $(find $O/prof -name "*kernel_stats.csv" | head -1) >> $O/log.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Name"]
    if "k_ri_" in n or "k_ploc_nn" in n:
        print("%-40s calls %5s total %9.3f ms avg %9.1f us" % (n.split("(anonymous namespace)::")[-1].split("(")[0], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
cat $O/log.txt
