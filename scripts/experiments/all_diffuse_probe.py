"""k_shade on the bench scene with EVERY material replaced by one diffuse record: the generic kernel vs a build whose BSDF
switch is compiled down to the diffuse case (-DGSP_EXPERIMENT_ONLY_DIFFUSE): what a per-class kernel of the diffuse class could gain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import gpuspectral_amd as g
from gpuspectral_amd import scenes, abi
sc = scenes.interior(1_000_000)
keep = sc.instances["emission"].sum(1) > 0
sc.instances["bsdf"] = np.where(keep, sc.instances["bsdf"], abi.bsdf_handle(abi.BSDF_DIFFUSE, 0))
sc.instances["bsdf"][keep] = abi.bsdf_handle(abi.BSDF_DIFFUSE, 0)
with g.Context(0) as ctx:
    ctx.upload_scene(sc); ctx.frame_begin(1920, 1080); ctx.render(spp=8); ts = 8
    for rep in range(2):
        ctx.reset_stats(); t = time.time(); ctx.render(spp=48, first_timestamp=ts, collect_kernel_times=1); ctx.sync(); dt = time.time() - t; ts += 48
        st = ctx.stats()
        print("%.1f Mrays/s | extend %.1f shade %.1f connect %.1f ms | vertices %d" % (st["traced_rays"] / dt / 1e6, st["extend_kernel_ms"], st["shade_kernel_ms"], st["connect_kernel_ms"], st["shaded_vertices"]), flush=True)
