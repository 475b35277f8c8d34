#!/bin/bash
# On the GPU box: which hardware VALU class counter (SQ_INSTS_VALU_ADD_F32 / MUL_F32 / FMA_F32 / TRANS_F32 / INT32 / INT64 / CVT)
# does each instruction kind of scripts/microbench/valu_rate.hip tick?  One rocprofv3 --pmc pass over the microbenchmark;
# every kernel of it runs ONE instruction kind, so counter / SQ_INSTS_VALU per kernel is that kind's class membership.
# Output: gpurun_out/<tag>/valu_class_calib.txt  (kernel, share of its VALU instructions in each class, measured cycles)
set -u
TAG=${1:-valu_class}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate $ROOT/scripts/microbench/valu_rate.hip > $OUT/build.log 2>&1 || { cat $OUT/build.log; exit 1; }
/tmp/valu_rate > $OUT/valu_rate.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT \
  --kernel-trace --output-format csv -d $OUT/pmc -- /tmp/valu_rate > $OUT/pmc.log 2>&1 || echo "pmc pass failed"
python3 - $OUT <<'PY' > $OUT/valu_class_calib.txt
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
per = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(root, "pmc", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        per[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
cls = ["ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT"]
print("%-16s %14s " % ("kernel", "SQ_INSTS_VALU") + " ".join("%9s" % c for c in cls) + "   (share of the kernel's VALU instructions)")
for k in sorted(per):
    tot = per[k].get("SQ_INSTS_VALU", 0.0)
    if tot <= 0: continue
    print("%-16s %14.0f " % (k, tot) + " ".join("%9.3f" % (per[k].get("SQ_INSTS_VALU_" + c, 0.0) / tot) for c in cls))
PY
find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/valu_class_calib.txt
