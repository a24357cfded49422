import os, sys, ctypes, numpy as np
REPO="/root/repo"
sys.path.insert(0, REPO)
# load the diag library in place of the product one
import importlib.util
p=os.path.join(REPO,"raytracer-utah_amd","__init__.py")
src=open(p).read().replace('hip = _load("librtu_hip.so")','hip = _load("librtu_hip_diag.so")')
spec=importlib.util.spec_from_loader("rtu_diag", loader=None)
m=importlib.util.module_from_spec(spec); m.__file__=p
exec(compile(src,p,"exec"), m.__dict__)
pkg=m
tag=sys.argv[1] if len(sys.argv)>1 else "teapot2_1080"
sc=pkg.Scene.from_blob_file(os.path.join(REPO,"tests/golden",tag,"scene.rtus.gz"))
W,H=sc.desc.camera.img_width, sc.desc.camera.img_height
ctx=pkg.Context(0); ctx.upload(sc)
fr=pkg.frame_setup(sc.desc.camera,W,H)
for _ in range(3): ctx.render(fr)
pkg.hip.rtu_diag_counters.restype=ctypes.c_void_p; pkg.hip.rtu_diag_counters.argtypes=[ctypes.c_void_p]
d=pkg.hip.rtu_diag_counters(ctx._h)
n=16+8*6*16384
buf=np.zeros(n,np.uint64)
pkg.hip.rtu_copy_to_host(ctx._h, buf.ctypes.data, d, n*8)
rec=buf[16:].reshape(8,16384,6).astype(np.float64)
for ph in range(1,5):
    r=rec[ph]; r=r[r[:,0]>0]
    busy=r[r[:,1]>0]
    if len(busy)==0: continue
    i=np.argmax(busy[:,0])
    print("phase",ph,"waves with work",len(busy),"R",busy[0,5],"max cycles %.0f"%busy[:,0].max(),"-> that wave: inner iters %d leaf iters %d tris %d t_inner %.0f"%(busy[i,1],busy[i,2],busy[i,3],busy[i,4]))
    print("   cycles per inner iter (t_inner/w_inner) median %.0f ; mean wave cycles %.0f; p99 %.0f"%(np.median(busy[:,4]/np.maximum(busy[:,1],1)), busy[:,0].mean(), np.percentile(busy[:,0],99)))
    tot=busy[:,0]; ti=busy[:,4]
    print("   inner share of wave time: %.2f ; leaf part per tri: %.0f cycles"%((ti.sum()/tot.sum()), ((tot-ti).sum()/np.maximum(busy[:,3].sum(),1))))
