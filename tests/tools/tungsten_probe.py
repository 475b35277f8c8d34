"""Renders the reference's shipped scenes at their film resolution on the GPU and writes 8x8 box-filtered linear
images next to the Tungsten fixtures' grid (gpurun_out/tungsten/): the data the region tolerances of
tests/test_gpu_reference_images.py were chosen from."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import abi

SPP = int(os.environ.get("SPP", "1024"))
out_dir = os.path.join(ROOT, "gpurun_out", "tungsten")
os.makedirs(out_dir, exist_ok=True)
FILM = {"cornell-box": (1024, 1024), "staircase2": (1024, 1024), "coffee": (800, 1000)}
with g.Context(0) as ctx:
    for name in sys.argv[1:] or list(FILM):
        W, H = FILM[name]
        sc = abi.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "ref_scenes", name + ".npz"))
        ctx.upload_scene(sc)
        ctx.frame_begin(W, H)
        t = time.time(); ctx.render(spp=SPP); img = ctx.download(); dt = time.time() - t
        lin = img[..., :3].astype(np.float64).reshape(H // 8, 8, W // 8, 8, 3).mean(axis=(1, 3))
        np.save(os.path.join(out_dir, "%s_%dspp.npy" % (name, SPP)), lin.astype(np.float32))
        print(json.dumps(dict(scene=name, spp=SPP, seconds=dt, mean=float(lin.mean()))), flush=True)
