#!/bin/bash
# One SQ counter pass over ab_probe for a given library: scripts/pmc_quick.sh tag lib.so
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcq_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GSP_LIB_PATH=$ROOT/$2
timeout 75 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/a -- python3 $ROOT/scripts/ab_probe.py > $OUT/a.log 2>&1 || echo pass a failed
timeout 75 rocprofv3 --pmc SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/b -- python3 $ROOT/scripts/ab_probe.py > $OUT/b.log 2>&1 || echo pass b failed
python3 - <<PY
import csv,glob,collections
for ps in ("a","b"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv"%ps):
        seen=set()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "k_trace" not in k: continue
            name="fused" if "FusedIO" in k else "extend" if "ExtendIO" in k else "connect"
            agg[name][r["Counter_Name"]]+=float(r["Counter_Value"])
            if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); cnt[name]+=1
    for n in agg:
        print("$1",ps,n,cnt[n],"launches:"," ".join("%s=%.3g"%(c,v/cnt[n]) for c,v in sorted(agg[n].items())))
PY
