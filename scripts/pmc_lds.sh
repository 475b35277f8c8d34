#!/bin/bash
# LDS counters of the traversal kernels over scripts/ab_probe.py for the current library and the named variants:
# scripts/pmc_lds.sh OUT variant...   (SQ_LDS_BANK_CONFLICT vs SQ_ACTIVE_INST_LDS vs SQ_INSTS_LDS per launch)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/$1; shift
: > $OUT
cd /tmp && export TMPDIR=/tmp
for v in current "$@"; do
  D=/tmp/pmclds_$v; rm -rf $D
  if [ $v != current ]; then export GSP_LIB_PATH=$ROOT/gpuspectral_amd/lib/variants/$v.so; else unset GSP_LIB_PATH; fi
  REPS=1 timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $D -- python3 $ROOT/scripts/ab_probe.py > $D.log 2>&1 || echo "pass $v failed" >> $OUT
  python3 - $v $D >> $OUT <<'PY'
import csv, glob, collections, sys
v, d = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        name = "extend" if "ExtendIO" in k else "connect" if "ConnectIO" in k else "shade" if "k_shade" in k else None
        if not name: continue
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name].add(r["Dispatch_Id"])
for n in sorted(agg):
    print(v, n, len(cnt[n]), "launches:", " ".join("%s=%.4g" % (c, x / len(cnt[n])) for c, x in sorted(agg[n].items())))
PY
done
cat $OUT
