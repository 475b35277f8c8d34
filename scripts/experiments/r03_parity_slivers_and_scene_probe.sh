mkdir -p gpurun_out/r03_w; O=gpurun_out/r03_w
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "living or path_rays or soup or deep_tree" 2>&1 | tail -3 > $O/log.txt
SECONDS=2 timeout 900 python tests/tools/scene_probe.py coffee staircase2 living-room interior caustics > $O/probe.txt 2>&1
python - >> $O/log.txt <<PY
import json
for l in open("$O/probe.txt"):
    if l.startswith("{"):
        d=json.loads(l); print("  %-11s %7d tris %6d nodes | ext %5.2f nodes %4.2f tris | shadow %5.2f / %4.2f | %7.1f Mrays/s %7.1f Msamples/s"%(d["scene"],d["triangles"],d["bvh_nodes"],d["nodes_per_ray"],d["tris_per_ray"],d["shadow_nodes_per_ray"],d["shadow_tris_per_ray"],d["mrays_per_s"],d["msamples_per_s"]))
PY
cat $O/log.txt
