mkdir -p gpurun_out/r05_o; O=gpurun_out/r05_o
GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/balanced.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed" > $O/parity_balanced.txt
for r in 1 2 3; do for v in current balanced chunk512 chunk4096; do echo -n "$v: "; if [ $v = current ]; then python scripts/ab_probe.py; else GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so python scripts/ab_probe.py; fi; done; done > $O/ab.txt 2>&1
for v in current balanced chunk512 chunk4096; do echo "== $v"; if [ $v = current ]; then SECONDS=2 python tests/tools/scene_probe.py coffee materials cornell-box staircase2; else GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so SECONDS=2 python tests/tools/scene_probe.py coffee materials cornell-box staircase2; fi; done > $O/scenes.txt 2>&1
cat $O/parity_balanced.txt $O/ab.txt
python - <<PY
import json
for l in open("$O/scenes.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print("  %-12s %8.1f Mrays/s  ns/vertex %.4f" % (d["scene"], d["mrays_per_s"], d["ns_per_vertex"]))
PY
