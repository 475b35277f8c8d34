cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05_prof
for mode in camera transform; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r05_prof/prof_$mode -- python3 $R/scripts/experiments/r05_edit_frame_breakdown.py 1920 1080 $mode device > $R/gpurun_out/r05_prof/prof_$mode.log 2>&1
  f=$(find $R/gpurun_out/r05_prof/prof_$mode -name "*kernel_stats.csv" | head -1)
  echo "== $mode"; grep -v "^E2\|^W2" $R/gpurun_out/r05_prof/prof_$mode.log | tail -1
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    n=r['Name']
    n=n.replace('gsp::(anonymous namespace)::','').replace('void gsp::','')[:70]
    print("%-72s calls %5s total %9.3f ms avg %9.1f us" % (n, r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
  find $R/gpurun_out/r05_prof/prof_$mode -name "*kernel_trace.csv" -delete
done
