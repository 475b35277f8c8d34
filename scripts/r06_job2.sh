# r06: phase budget of k_trace -- wave profile + ablation by padding (what an issue slot of each phase is worth), one box
set -u
O=gpurun_out/r06b; mkdir -p $O
V=$PWD/gpuspectral_amd/lib/variants
GSP_LIB_PATH=$V/waveprof.so timeout 300 python scripts/trace_phase_budget.py gpu $O/wave_profile.json > $O/wave_profile.log 2>&1; echo "waveprof rc $?"
: > $O/ab_padding.txt
for round in 1 2 3; do
  echo -n "current: " >> $O/ab_padding.txt; timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/ab_padding.txt
  for v in padnode32 padleaf32 padbook16; do
    echo -n "$v: " >> $O/ab_padding.txt; GSP_LIB_PATH=$V/$v.so timeout 300 python scripts/ab_probe.py 2>&1 | tail -1 >> $O/ab_padding.txt
  done
done
cat $O/ab_padding.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06b/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['msamples_per_s'], d['ms_per_step'], d['config'].get('download_ms'), d['roofline']['bound'], d['roofline']['frac'], d['roofline']['valu_instr_per_ray'], d['roofline']['valu_lane_instr_per_ray'])
o=d['roofline']['other_kernels']['k_trace<ConnectIO>']
print({k:o.get(k) for k in ('issue_frac','lanes_per_instr')})
PY
