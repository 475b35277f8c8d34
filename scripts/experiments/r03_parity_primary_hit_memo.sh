mkdir -p gpurun_out/r03_h; O=gpurun_out/r03_h
echo "== parity with the primary-hit memo" > $O/log.txt
timeout 1500 python -m pytest tests/ -m gpu -x -q 2>&1 | tail -5 >> $O/log.txt
for i in 1 2 3; do
echo -n "memo on : " >> $O/log.txt; timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/log.txt
echo -n "memo off: " >> $O/log.txt; GSP_PRIMARY_MEMO=0 timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/log.txt
done
timeout 600 python bench.py --no-cpu-baseline > $O/bench_memo.json 2>$O/bench_memo.err
GSP_PRIMARY_MEMO=0 timeout 600 python bench.py --no-cpu-baseline > $O/bench_nomemo.json 2>$O/bench_nomemo.err
cat $O/log.txt; python - <<PY
import json
for n in ("memo","nomemo"):
    try:
        b=json.loads(open("$O/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, "value %.1f Mrays/s, %.2f Msamples/s, ms/step %.1f, extend %.1f shade %.1f connect %.1f, memoised %d"%(b["value"], b["msamples_per_s"], b["ms_per_step"], b["roofline"]["extend_ms"], b["roofline"]["shade_ms"], b["roofline"]["connect_ms"], b["config"].get("memoised_rays",0)))
    except Exception as e: print(n, "failed", e)
PY
