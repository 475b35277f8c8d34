set -u
mkdir -p gpurun_out/r06a
export GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/waveprof.so
timeout 300 python scripts/trace_phase_budget.py gpu gpurun_out/r06a/wave_profile.json > gpurun_out/r06a/wave_profile.log 2>&1; echo "waveprof rc $?"
unset GSP_LIB_PATH
timeout 600 python tests/tools/share_probe.py --equal-spp > gpurun_out/r06a/share_probe.txt 2>&1; echo "share rc $?"
timeout 600 python bench.py > gpurun_out/r06a/bench.json 2> gpurun_out/r06a/bench.err; echo "bench rc $?"
tail -c 1500 gpurun_out/r06a/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06a/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['msamples_per_s'], d['ms_per_step'], d['config'].get('download_ms'), d['roofline']['bound'], d['roofline']['frac'], d['roofline']['pmc'])
for w in d['config'].get('other_workloads',[]): print(w)
PY
