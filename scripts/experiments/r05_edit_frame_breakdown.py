"""Where does a frame with a transform edit spend its time?  (4th argument `device`: the frame is shown from device memory.)  500x500, one sample per frame, the 988 k-triangle scene: wall time of
gsp_update_instances / gsp_render / gsp_peek per frame, for a transform edit, a camera edit and no edit.
   python scripts/experiments/r05_edit_frame_breakdown.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import scenes

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (500, 500)
DEVICE_PEEK = len(sys.argv) > 4 and sys.argv[4] == "device"  # gsp_peek_to_device instead of gsp_peek
if DEVICE_PEEK:
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    dptr = C.c_void_p()
    assert hip.hipMalloc(C.byref(dptr), C.c_size_t(W * H * 16)) == 0
sc = scenes.interior(1_000_000, seed=7)
LANES = int(sys.argv[5]) if len(sys.argv) > 5 else 1  # 5th argument: pipeline lanes of the context (gsp_ctx_options.lanes)
from gpuspectral_amd import abi
with g.Context(0, options=abi.CtxOptions(lanes=LANES) if LANES != 1 else None) as ctx:
    ctx.upload_scene(sc)
    for what in (sys.argv[3].split(",") if len(sys.argv) > 3 else ("none", "camera", "transform")):
        ctx.frame_begin(W, H)
        ctx.render(spp=4)
        ctx.sync()
        if what == "transform":  # the first edits of a scene are one-offs (refit in place, then the scene is built as two trees): timed apart
            for w in range(2):
                i = sc.instances.copy()
                t = i["transform"][len(i) - 1].copy()
                t[13] += np.float32(0.0005)
                i["transform"][len(i) - 1] = t
                sc.instances = i
                t0 = time.time()
                ctx.update_instances(i)
                print("          one-off edit %d: %.2f ms" % (w, 1e3 * (time.time() - t0)))
                ctx.render(spp=1, first_timestamp=1000 + w)
        tt = np.zeros(3)
        edits = []
        n = 60
        st0 = ctx.stats()
        t_all = time.time()
        for f in range(n):
            t0 = time.time()
            if what == "transform":
                i = sc.instances.copy()
                t = i["transform"][len(i) - 1].copy()
                t[13] += np.float32(0.0005)
                i["transform"][len(i) - 1] = t
                sc.instances = i
                ctx.update_instances(i)
            elif what == "camera":
                m = np.array(sc.to_world, np.float32).copy()
                m[12] += np.float32(0.002 * f)
                ctx.update_camera(m, sc.fov)
            t1 = time.time()
            ctx.render(spp=1, first_timestamp=4 + f)
            t2 = time.time()
            if DEVICE_PEEK:
                ctx.peek_to_device(dptr.value, W * H * 16)
            else:
                ctx.peek()
            t3 = time.time()
            tt += (t1 - t0, t2 - t1, t3 - t2)
            edits.append(1e3 * (t1 - t0))
        ctx.sync()
        el = time.time() - t_all
        st = ctx.stats()
        rays = st["extension_rays"] + st["shadow_rays"] - st0["extension_rays"] - st0["shadow_rays"]
        print("%-9s %dx%d: %7.1f frames/s | per frame: edit %.3f ms, render %.3f ms, peek %.3f ms | %.1f Mrays/s | updates %d refits %d drains %d, %.1f GB held" %
              (what, W, H, n / el, *(1e3 * tt / n), rays / el / 1e6, st["scene_updates"], st["scene_refits"], st["scene_drains"], st["device_bytes"] / 1e9) + (", splits %d" % st["scene_splits"] if st.get("scene_splits") else ""))
        if what == "transform":
            print("          edit call, ms: first %.2f, median %.3f, max of the rest %.3f" % (edits[0], float(np.median(edits)), max(edits[1:])))
