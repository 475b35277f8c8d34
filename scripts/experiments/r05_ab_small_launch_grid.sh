cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in sw_cur sw_rpl8 sw_rpl16 sw_rpl32; do
  export GSP_LIB_PATH=$R/gpuspectral_amd/lib/variants/$v.so
  for res in "500 500" "1000 800"; do
    rm -rf /tmp/prof_$v
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$v -- python3 $R/scripts/experiments/r05_edit_frame_breakdown.py $res none device > /tmp/prof_$v.log 2>&1
    f=$(find /tmp/prof_$v -name "*kernel_stats.csv" | head -1)
    fps=$(grep -o "[0-9.]* frames/s" /tmp/prof_$v.log | head -1)
    python3 - "$f" "$v" "$res" "$fps" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
d={}
for r in rows:
    n=r['Name']
    for k,t in (('ExtendIO','extend'),('ConnectIO','connect'),('k_shade<false, false>','shade')):
        if k in n and 'MemoIO' not in n: d[t]=float(r['AverageNs'])/1e3
print("%-9s %-9s %s | extend %.1f us connect %.1f us shade %.1f us" % (sys.argv[2], sys.argv[3], sys.argv[4], d.get('extend',0), d.get('connect',0), d.get('shade',0)))
PY
  done
done
