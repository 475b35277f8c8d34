"""Bias attribution on the reference's 'Modern Hall' scene (staircase2: rough plastic, twosided, 16 + 160 triangle lights with the
built-in disks), CPU oracle against the Tungsten fixture: which of the reference's departures (oracle/oracle_bsdf.h kQuirk*) costs what
when a scene has MANY lights.   python tests/tools/quirk_probe_staircase2.py [size] [spp]  > profiles/r06_quirk_attribution_staircase2.txt
Not a convergence test: the BSDF models differ from Tungsten's (Beckmann-sampled GGX-valued conductor, the plastic's coupling term) and
the loader skips `sphere` geometry; the Cornell box (diffuse only) is the clean case (tests/tools/quirk_probe.py)."""
import os, sys, time, tarfile, tempfile
import numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
from conftest import GOLDEN
from test_gpu_reference_images import _cells
from gpuspectral_amd import host
from oracle import oracle as orc
d=tempfile.mkdtemp()
for t in ("staircase2.tar.xz","staircase2_textures.tar"):
    with tarfile.open(os.path.join(GOLDEN,"ref_scenes",t)) as f: f.extractall(d)
xml=os.path.join(d,"staircase2","scene.xml")
fix=np.load(os.path.join(GOLDEN,"ref_scenes","tungsten_staircase2.npz"))
size=int(sys.argv[1]) if len(sys.argv)>1 else 256
spp=int(sys.argv[2]) if len(sys.argv)>2 else 32
L=orc.lib()
for dorm in (True,):
    sc=host.Scene(xml, dormant_features=dorm, builtin_shapes=True).arrays()
    for mask in (0,15,11,4):
        L.oracle_set_quirks_off(mask)
        t0=time.time()
        o=orc.Oracle(sc); img,st=o.render(size,size,spp=spp); o.close()
        L.oracle_set_quirks_off(0)
        k=size//128
        ours=img[:,:3].astype(np.float64).reshape(128,k,128,k,3).mean(axis=(1,3))
        for strict in (False,True):
            r,_=_cells(ours,fix,strict=strict)
            v=r[np.isfinite(r)]
            print("dormant=%s quirks_off=%2d strict=%d cells %3d: min %.3f max %.3f median %.3f mean|log2| %.3f  (%.0fs)"%(dorm,mask,strict,v.size,v.min(),v.max(),np.median(v),np.abs(np.log2(v)).mean(),time.time()-t0))
        if mask in (0,15):
            g=np.nanmean(_cells(ours,fix,strict=True)[0],axis=2)
            for row in g: print("   "+" ".join("  -  " if not np.isfinite(x) else "%5.2f"%x for x in row))
