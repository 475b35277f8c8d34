"""Cost of the per-frame scene edits (ABI 5) in the reference's own usage: a 500x500 window, one sample per frame
(S/main.cpp:17, Renderer::run), with the scene edited between frames the way a viewer does -- and the latency of each update call
on the 1 M-triangle bench scene.   python scripts/update_latency_probe.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import scenes


def moved(m, k):
    m = np.array(m, np.float32).copy()
    m[12] += np.float32(0.002 * k)
    return m


def loop(ctx, sc, frames, edit):
    """frames x {edit, +1 spp, peek}; -> frames per second"""
    ctx.frame_begin(500, 500)
    ctx.render(spp=4)
    ctx.sync()
    t = time.time()
    for f in range(frames):
        edit(f)
        ctx.render(spp=1, first_timestamp=4 + f)
        ctx.peek()
    ctx.sync()
    return frames / (time.time() - t)


with g.Context(0) as ctx:
    for name, sc in (("cornell (36 tris)", scenes.cornell_materials(0) if False else None), ("interior (988 k tris)", scenes.interior(1_000_000, seed=7))):
        if sc is None:
            from gpuspectral_amd import host
            sc = host.Scene(os.path.join(ROOT, "tests", "golden", "cornell-box", "scene.xml")).arrays()
        ctx.upload_scene(sc)
        st = ctx.stats()
        print("== %s: upload + build %.1f ms" % (name, st["bvh_build_ms"]))
        print("   no edits                  : %7.1f frames/s" % loop(ctx, sc, 300, lambda f: None))
        def noop(f):
            ctx.update_tables(sc); ctx.update_instances(sc.instances); ctx.update_camera(sc.to_world, sc.fov)
        print("   all three calls, no change: %7.1f frames/s (what a host that re-reads everything per frame pays)" % loop(ctx, sc, 300, noop))
        print("   camera moves every frame  : %7.1f frames/s" % loop(ctx, sc, 300, lambda f: ctx.update_camera(moved(sc.to_world, f), sc.fov)))
        def tables(f):
            b = [x.copy() for x in sc.bsdfs]
            if len(b[0]):
                b[0]["reflectance"][0] = (0.2 + 0.001 * (f % 100), 0.3, 0.4)
            sc.bsdfs = b
            ctx.update_tables(sc)
        print("   a BSDF record every frame : %7.1f frames/s" % loop(ctx, sc, 300, tables))
        def inst(f):
            i = sc.instances.copy()
            t = i["transform"][len(i) - 1].copy()
            t[13] += np.float32(0.0005)
            i["transform"][len(i) - 1] = t
            sc.instances = i
            ctx.update_instances(i)
        n = 300  # (r05: an edit of one object refits a small tree: the 988 k scene affords as many frames as Cornell)
        print("   a transform every frame   : %7.1f frames/s (the scene is built as two trees at the second edit, then the edited object's small tree is refitted into the next slot of its ring each frame; Cornell is too small to split: whole trees through the ring; ABI 5 rebuilt the tree: 29.6 on the 988 k scene)" % loop(ctx, sc, n, inst))
        st = ctx.stats()
        print("   (edits that changed something %d, refits %d, edits that first let the samples in flight finish %d, scene built as two trees %d times; %.1f GB held)" %
              (st["scene_updates"], st["scene_refits"], st["scene_drains"], st["scene_splits"], st["device_bytes"] / 1e9))
        # bare latencies with an idle pipeline
        for what, fn in (("gsp_update_camera", lambda k: ctx.update_camera(moved(sc.to_world, 1000 + k), sc.fov)), ("gsp_update_tables", lambda k: tables(1000 + k)),
                         ("gsp_update_instances", lambda k: inst(k))):
            ctx.sync()
            ts = []
            for k in range(8):
                t = time.time(); fn(k); ts.append(time.time() - t)
            print("   %-22s %8.3f ms (median of 8, idle pipeline)" % (what, 1e3 * sorted(ts)[4]))
