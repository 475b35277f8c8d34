#!/bin/bash
# one guarded rocprofv3 --pmc pass: scripts/profile_one.sh <tag> "<counters>" [bench args]
TAG=$1; CNT=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 100 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/p -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline $* > $OUT/p.log 2>&1
echo "rc=$?"
python3 - <<PY
import csv,glob,collections
per=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]; k=("k_trace" if "k_trace" in k else "k_shade" if "k_shade" in k else None)
        if not k: continue
        per[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k in per: print(k, dict(per[k]))
PY
