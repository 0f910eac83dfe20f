import sys, time, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from oracle import host_prep as hp, oracle as orc
from pathtracer_amd.device import PathTracer
import pickle
scenes = pickle.load(open('tests/golden/_scenes_tmp.pkl','rb'))
pt = PathTracer(0)
# KAT math
rng = np.random.default_rng(1)
x = (rng.random(100000, dtype=np.float32)*40-20).astype(np.float32)
y = (rng.random(100000, dtype=np.float32)*40-20).astype(np.float32)
for op,(a,b) in {'sin':(x,None),'cos':(x,None),'asin':(x/20,None),'acos':(x/20,None),'log':(np.abs(x)+1e-30,None),'log2':(np.abs(x),None),'exp2':(x*6,None),'atan2':(x,y),'pow':(np.abs(x),y/4),'min':(x,y),'max':(x,y),'sqrt':(np.abs(x),None),'div':(x,y),'fma_aba':(x,y)}.items():
    g = pt.kat_math(op,a,b); o = orc.math_op(op,a,b)
    bad = (g.view(np.uint32)!=o.view(np.uint32)) & ~(np.isnan(g)&np.isnan(o))
    print('kat', op, 'mismatch', int(bad.sum()))
z = np.array([0.0,-0.0,0.0,-0.0,np.nan,1.0],dtype=np.float32); w=np.array([-0.0,0.0,0.0,-0.0,1.0,np.nan],dtype=np.float32)
for op in ('min','max'):
    print(op, pt.kat_math(op,z,w).view(np.uint32), orc.math_op(op,z,w).view(np.uint32))
for name,(sb,W,H,n) in scenes.items():
    cam = hp.camera_to_gpu(sb.camera, W, H)
    sc = orc.OracleScene(sb)
    ref = orc.render(sc, cam, n, want_radiance=True, counters=True)
    pt.upload_scene(sb.figs, sb.mats, sb.nodes, sb.prims, sb.lights, sb.infinite, sb.tex_pixels, sb.tex_infos)
    pt.resize(W,H); pt.debug_capture(n)
    t=time.time(); pt.trace_n(cam, 0, 1, n); pt.sync(); dt=time.time()-t
    ids = pt.read_primary_ids(); acc = pt.read_accum(); out = pt.read_output(); rad = pt.read_debug_radiance(n); st = pt.stats()
    print(name, 'gpu %.3fs'%dt, 'ids mismatch', int((ids!=ref['primary_id']).sum()), 'rad mismatch', int((rad.view(np.uint32)!=ref['radiance'].view(np.uint32)).sum()), 'of', rad.size,
          'accum mismatch', int((acc.view(np.uint32)!=ref['accum'].view(np.uint32)).sum()), 'rgba8 mismatch', int((out!=ref['rgba8']).sum()), st, ref['counters']['paths'], ref['counters']['bounces'], ref['counters']['shadow_rays'])
    bad = np.argwhere((rad.view(np.uint32)!=ref['radiance'].view(np.uint32)).any(axis=-1))
    if len(bad): print('  first bad', bad[:5], rad[tuple(bad[0])], ref['radiance'][tuple(bad[0])])
    # single-step equivalence
    pt.clear(); 
    for k in range(n):
        c2 = cam.copy(); c2[0]['frame_index']=k; c2[0]['sample_count']=k+1
        pt.trace(c2)
    acc2 = pt.read_accum()
    print('   n x trace vs trace_n mismatch', int((acc2.view(np.uint32)!=acc.view(np.uint32)).sum()))
    pt.debug_capture(0)
