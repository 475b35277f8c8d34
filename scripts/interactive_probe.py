"""The reference's usage pattern (main.cpp:20-28): one sample per frame at 500x500, the image shown every frame.
Frames per second with gsp_download (waits for the stragglers) and with gsp_peek (shows what is folded)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes
for name, sc in (("cornell_materials(48)", scenes.cornell_materials(48)), ("interior(600k)", scenes.interior(600_000))):
    with g.Context(0) as ctx:
        ctx.upload_scene(sc)
        for mode in ("download", "peek"):
            ctx.frame_begin(500, 500)
            ctx.render(spp=1); ctx.sync()
            n = 200
            t = time.time()
            for f in range(n):
                ctx.render(spp=1, first_timestamp=1 + f)
                if mode == "download":
                    ctx.download_compact()
                else:
                    img, k = ctx.peek()
            shown = 1 + n if mode == "download" else k
            dt = time.time() - t
            ctx.sync()
            print("%-22s %-8s: %6.1f frames/s (%.2f ms per frame), image %d of %d samples behind at the last frame" % (
                name, mode, n / dt, dt / n * 1e3, 1 + n - shown, 1 + n), flush=True)
