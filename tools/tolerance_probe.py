import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np
from pathtracer_amd import host
from pathtracer_amd.device import PathTracer
sc = host.Scene.from_fixture('/root/repo/data/scenes/fractals.npz').build()
pt = PathTracer(0)
sc.upload_to(pt)
w,h,spp=1920,1080,128
pt.resize(w,h)
cam=sc.camera(w,h)
LUM=np.array([0.2126,0.7152,0.0722])
res={}
for mode in (0,1,2):
    pt.set_option("arith_mode", mode)
    pt.clear(); pt.trace_n(cam,0,1,spp); pt.sync()
    pt.clear()
    t0=time.perf_counter()
    for i in range(3): pt.trace_n(cam,i*spp,1+i*spp,spp)
    pt.sync(); dt=time.perf_counter()-t0
    a=pt.read_accum()[...,:3].astype(np.float64); ids=pt.read_primary_ids()
    res[mode]=(a,ids)
    print("mode",mode,"Mpaths/s",round(w*h*spp*3/dt/1e6,1), flush=True)
for mode in (1,2):
    g,r=res[mode][0],res[0][0]
    print("mode",mode,"rel rmse vs exact at 384 spp", float(np.sqrt((((g-r)@LUM)**2).mean())/(r@LUM).mean()), "ids differing", int((res[mode][1]!=res[0][1]).sum()), "mean lum ratio", float((g@LUM).mean()/(r@LUM).mean()))
