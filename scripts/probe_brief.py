"""stdin: JSON lines of tests/tools/scene_probe.py -> one short line per scene (rate, per-kernel ns per unit of work)."""
import json, sys
for l in sys.stdin:
    if not l.startswith("{"):
        print(l.rstrip()); continue
    d = json.loads(l)
    print("%-12s %8.1f Mrays/s | ns/ext-ray %.4f  ns/vertex %.4f  ns/shadow-ray %.4f | ext %.0f shade %.0f conn %.0f ms (%d spp)" % (
        d["scene"], d["mrays_per_s"], d["ns_per_ext_ray"], d["ns_per_vertex"], d["ns_per_shadow_ray"], d["extend_ms"], d["shade_ms"], d["connect_ms"], d["spp"]))
