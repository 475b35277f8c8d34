"""Debug build only (-DGSP_WAVE_PROFILE): lane occupancy of the closest-hit traversal loop."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpuspectral_amd as g
from gpuspectral_amd import scenes
sc = scenes.interior(1_000_000)
with g.Context(0) as ctx:
    ctx.upload_scene(sc); ctx.frame_begin(1920, 1080); ctx.render(spp=16); ctx.sync()
    L = ctypes.CDLL(os.environ["GSP_LIB_PATH"])
    out = (ctypes.c_ulonglong * 16)()
    L.gsp_debug_wave_profile(out)
    ns, nl, ls, ll, loops, rf, rl, idle, stall = [out[i] for i in range(9)]
    print("end game (hand-out empty): %.1f %% of the node steps, %.1f lanes enabled per step" % (100.0 * out[9] / ns, out[10] / max(out[9], 1)))
    print("node steps %d, lanes/step %.1f (idle %.1f, stalled on a leaf %.1f) | leaf steps %d, lanes/step %.1f | node:leaf steps %.2f | loop passes %d | refills %d, lanes/refill %.1f" % (
        ns, nl / ns, idle / ns, stall / ns, ls, ll / ls, ns / ls, loops, rf, rl / max(rf, 1)))
    st = ctx.stats()
    print("rays %d: node steps x64 per ray %.1f, leaf steps x64 per ray %.1f" % (st["extension_rays"], ns * 64 / st["extension_rays"], ls * 64 / st["extension_rays"]))
