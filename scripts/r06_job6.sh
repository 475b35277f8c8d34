set -u
O=gpurun_out/r06e; mkdir -p $O
timeout 300 python tests/tools/download_probe.py > $O/download_probe.txt 2>&1; cat $O/download_probe.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scene_updates.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06e/bench_s20.json').read().strip().splitlines()[-1])
print(d['value'], d['msamples_per_s'], d['ms_per_step'], d['config']['download_ms'], d['roofline']['bound'], d['roofline']['frac'], d['roofline']['pmc'])
PY
