"""k_finish takes over a drain below gsp_ctx_options.finish_paths live paths (default 262144, chosen in r03 when the kernel still
paid one atomic per path).  Re-scan: drained 500x500 one-sample frames per second (an edit before every frame forces the drain)
and the time of an 8-spp 1080p call + sync, on the Cornell box and the 988 k-triangle scene.   python scripts/finish_threshold_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuspectral_amd as g
from gpuspectral_amd import abi, scenes, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cornell = host.Scene(os.path.join(ROOT, "tests", "golden", "cornell-box", "scene.xml")).arrays()
interior = scenes.interior(1_000_000, seed=7)
print("# %-10s %-9s %16s %16s" % ("finish", "scene", "drained frames/s", "8 spp 1080p ms"))
for fp in (1 << 16, 1 << 18, 1 << 20, 1 << 22, 0xFFFFFFFF):
    for name, sc in (("cornell", cornell), ("interior", interior)):
        with g.Context(0, options=abi.CtxOptions(finish_paths=fp)) as ctx:
            ctx.upload_scene(sc)
            ctx.frame_begin(500, 500)
            ctx.render(spp=4); ctx.sync()
            t = time.time()
            for f in range(200):
                ctx.render(spp=1, first_timestamp=4 + f)
                ctx.sync()
            fps = 200 / (time.time() - t)
            ctx.frame_begin(1920, 1080)
            ctx.render(spp=8); ctx.sync()
            ts = []
            for k in range(5):
                t = time.time(); ctx.render(spp=8, first_timestamp=8 + 8 * k); ctx.sync(); ts.append(time.time() - t)
            print("  %-10s %-9s %16.1f %16.2f" % ("never" if fp == 0xFFFFFFFF else str(fp), name, fps, 1e3 * sorted(ts)[2]), flush=True)
