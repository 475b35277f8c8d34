"""gsp_download of the 1080p frame into a host buffer the caller owns: ms per call (r06: pinned double-buffered read-back)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import scenes
with g.Context(0) as ctx:
    ctx.upload_scene(scenes.cornell_materials(12))
    for (W, H) in ((1920, 1080), (4096, 4096)):
        ctx.frame_begin(W, H); ctx.render(spp=1); ctx.sync()
        out = np.ones((H, W, 4), np.float32)
        ts = []
        for _ in range(6):
            t = time.perf_counter(); ctx.download(out=out); ts.append((time.perf_counter() - t) * 1e3)
        t = time.perf_counter(); fresh = ctx.download(); t_fresh = (time.perf_counter() - t) * 1e3
        assert np.array_equal(fresh, out)
        print("%dx%d (%.0f MB): gsp_download into a resident buffer %s ms; into a fresh np.zeros buffer %.1f ms -> %.1f GB/s best"
              % (W, H, out.nbytes / 1e6, " ".join("%.1f" % x for x in ts), t_fresh, out.nbytes / 1e6 / min(ts)), flush=True)
