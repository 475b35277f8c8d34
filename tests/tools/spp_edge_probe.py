import sys; import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'oracle'))
import numpy as np, gpuspectral_amd as g, oracle as O
from gpuspectral_amd import scenes
sc=scenes.cornell_materials(8); o=O.Oracle(sc)
with g.Context(0) as c:
    c.upload_scene(sc)
    for (W,H,spp) in ((1,1,100000),(8,8,5000),(3,2,70000)):
        c.frame_begin(W,H); c.render(spp=spp); img=c.download().reshape(-1,4)
        ref,_=o.render(W,H,spp=spp)
        print(W,H,spp,"equal:",np.array_equal(img,ref))
