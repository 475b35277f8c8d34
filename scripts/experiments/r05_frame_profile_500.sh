# rocprofv3 kernel stats of the 500x500 one-sample-per-frame loop without edits: what is a frame made of?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mode=${1:-none}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_s/prof500_$mode -- python3 $R/scripts/experiments/r05_edit_frame_breakdown.py 500 500 $mode > $R/gpurun_out/r05_s/prof500_$mode.log 2>&1
f=$(find $R/gpurun_out/r05_s/prof500_$mode -name "*kernel_stats.csv" | head -1)
grep -v "^E2\|^W2" $R/gpurun_out/r05_s/prof500_$mode.log | tail -2
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    n=r['Name'].replace('gsp::(anonymous namespace)::','').replace('void gsp::','')[:70]
    print("%-72s calls %5s total %9.3f ms avg %9.1f us" % (n, r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
find $R/gpurun_out/r05_s/prof500_$mode -name "*kernel_trace.csv" -delete
