mkdir -p gpurun_out/r03_v; O=gpurun_out/r03_v
( echo "## default (parity collapse)"; SECONDS=2 timeout 900 python tests/tools/scene_probe.py coffee staircase2 interior caustics
  echo "## GSP_COLLAPSE=greedy"; GSP_COLLAPSE=greedy SECONDS=2 timeout 900 python tests/tools/scene_probe.py coffee staircase2 interior caustics
  echo "## GSP_BVH=lbvh"; GSP_BVH=lbvh SECONDS=2 timeout 900 python tests/tools/scene_probe.py coffee staircase2 interior caustics ) > $O/log.txt 2>&1
python - <<PY
import json
sec=None
for l in open("$O/log.txt"):
    if l.startswith("##"): sec=l.strip(); print(sec); continue
    if l.startswith("{"):
        d=json.loads(l); print("  %-11s %7d tris %6d nodes | ext %5.2f nodes %4.2f tris | shadow %5.2f / %4.2f | %7.1f Mrays/s %7.1f Msamples/s"%(d["scene"],d["triangles"],d["bvh_nodes"],d["nodes_per_ray"],d["tris_per_ray"],d["shadow_nodes_per_ray"],d["shadow_tris_per_ray"],d["mrays_per_s"],d["msamples_per_s"]))
PY
