mkdir -p gpurun_out/r03_t; O=gpurun_out/r03_t
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "deep_tree" 2>&1 | tail -12 > $O/log.txt
python - >> $O/log.txt 2>&1 <<PY
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import importlib.util
spec=importlib.util.spec_from_file_location("tp","tests/test_gpu_parity.py"); tp=importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
import gpuspectral_amd as g
for n in (60, 120, 220, 400):
    sc=tp._nested_scene(n)
    with g.Context(0) as ctx:
        ctx.upload_scene(sc); ctx.frame_begin(8,8); ctx.render(spp=1); st=ctx.stats()
        print("nested", n, "tris", st["num_triangles"], "depth", st["bvh_depth"], "nodes", st["num_bvh_nodes"])
from gpuspectral_amd import scenes
for name, sc in (("interior 1M", scenes.interior(1_000_000, seed=7)), ("caustics 1M", scenes.caustics(1_000_000, seed=11))):
    with g.Context(0) as ctx:
        ctx.upload_scene(sc); ctx.frame_begin(8,8); ctx.render(spp=1); st=ctx.stats()
        print(name, "depth", st["bvh_depth"], "nodes", st["num_bvh_nodes"])
PY
cat $O/log.txt
