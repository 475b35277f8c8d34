mkdir -p gpurun_out/r05_g; O=gpurun_out/r05_g
python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_configs.py tests/test_gpu_scene_updates.py tests/test_gpu_textures.py -m gpu -x -q 2>&1 | tail -3 > $O/parity.txt
GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/lazyshear.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2 > $O/parity_lazyshear.txt
for r in 1 2 3; do for v in r05b current lazyshear; do echo -n "$v: "; if [ $v = current ]; then python scripts/ab_probe.py; else GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so python scripts/ab_probe.py; fi; done; done > $O/ab.txt 2>&1
for v in r05b current lazyshear; do echo "== $v"; if [ $v = current ]; then SECONDS=2 python tests/tools/scene_probe.py coffee materials staircase2 cornell-box caustics; else GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/$v.so SECONDS=2 python tests/tools/scene_probe.py coffee materials staircase2 cornell-box caustics; fi; done > $O/scenes.txt 2>&1
cat $O/parity.txt $O/parity_lazyshear.txt $O/ab.txt
python - <<PY
import json
for l in open("$O/scenes.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print("  %-12s %8.1f Mrays/s  ns/vertex %.4f  ns/shadow ray %.4f  ns/ext ray %.4f" % (d["scene"], d["mrays_per_s"], d["ns_per_vertex"], d["ns_per_shadow_ray"], d["ns_per_ext_ray"]))
PY
