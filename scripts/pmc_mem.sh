#!/bin/bash
# (at most two counters of a block per pass: more and the profile cannot be built, and the run hangs until its timeout)
# Vector-memory path counters (TA / TCP / TD / address translation) of the traversal kernels over ab_probe: scripts/pmc_mem.sh tag [lib.so]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcm_$1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -n "$2" ] && export GSP_LIB_PATH=$ROOT/$2
P=1
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum"; do
  timeout 75 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$P -- python3 $ROOT/scripts/ab_probe.py > $OUT/p$P.log 2>&1 || echo pass $P failed
  P=$((P+1))
done
python3 - <<PY
import csv,glob,collections
for ps in ("p1","p2","p3","p4","p5"):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter(); dur=collections.Counter()
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv"%ps):
        seen=set()
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            name="extend" if "ExtendIO" in k else "connect" if "ConnectIO" in k else "shade" if "k_shade" in k else None
            if not name: continue
            agg[name][r["Counter_Name"]]+=float(r["Counter_Value"])
            if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); cnt[name]+=1
    for n in sorted(agg):
        print("$1",ps,n,cnt[n],"launches, per launch:"," ".join("%s=%.4g"%(c,v/cnt[n]) for c,v in sorted(agg[n].items())))
PY
