import os, sys
sys.path.insert(0, os.getcwd())
import gpuspectral_amd as g
from gpuspectral_amd import scenes
sc = scenes.interior(1_000_000)
with g.Context(0) as ctx:
    ctx.upload_scene(sc); ctx.frame_begin(1920,1080)
    ctx.render(spp=4)
    ctx.render(spp=16, first_timestamp=4, collect_kernel_times=2)  # 2: one line per iteration on stderr
