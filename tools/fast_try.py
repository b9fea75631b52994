import importlib, sys, time, numpy as np
sys.path.insert(0, "/root/repo")
rta=importlib.import_module("ray-tracing_amd"); scenes=importlib.import_module("ray-tracing_amd.scenes")
W=int(sys.argv[1]) if len(sys.argv)>1 else 512
which=sys.argv[2] if len(sys.argv)>2 else "battle"
spp=int(sys.argv[3]) if len(sys.argv)>3 else 64
out=getattr(scenes, which)(W=W); s,cam,flags=out[:3]
r=rta.Renderer(0); r.upload(s)
import ctypes
a_,b_=ctypes.c_int(),ctypes.c_int(); rta.lib.rt_debug_fast_occupancy(r._h, ctypes.byref(a_), ctypes.byref(b_)); print("fast kernel: waves per CU", a_.value, "LDS bytes", b_.value)
imgs={}
for name,fl in (("f64",0),("fast",rta.RT_F_FAST)):
    r.accum_clear(W,W)
    r.render(cam, rta.make_params(W,W,spp,flags=fl)); r.sync()
    r.accum_clear(W,W)
    t0=time.time(); r.render(cam, rta.make_params(W,W,spp,flags=fl)); r.sync(); dt=time.time()-t0
    ms,nl=r.last_render_ms()
    a=r.accum_download(W,W); img=r.resolve(W,W,spp)
    imgs[name]=(a,img)
    print(name, "%.1f Mpaths/s"%(W*W*spp/ms/1e3), "ms",ms, "nan frac", np.isnan(a).mean(), "mean", np.nanmean(a))
d=imgs["f64"][1].astype(float)-imgs["fast"][1].astype(float)
print("image diff: mean %.4f rmse %.3f"%(d.mean(), np.sqrt((d*d).mean())))
# same-scene noise floor: f64 with another seed
r.accum_clear(W,W); r.render(cam, rta.make_params(W,W,spp,seed=77,flags=0)); r.sync(); i2=r.resolve(W,W,spp).astype(float)
d2=imgs["f64"][1].astype(float)-i2
print("floor (f64 seed 10 vs 77): mean %.4f rmse %.3f"%(d2.mean(), np.sqrt((d2*d2).mean())))
rta.write_png("gpurun_out/fast_%s.png"%which, imgs["fast"][1]); rta.write_png("gpurun_out/f64_%s.png"%which, imgs["f64"][1])
a64,afast=imgs["f64"][0],imgs["fast"][0]
rel=(afast.sum(2)-a64.sum(2))/np.maximum(a64.sum(2),1e-9)
print("radiance ratio fast/f64 by image rows (top to bottom of accumulator):")
for r0 in range(0,W,W//8):
    blk64=a64[r0:r0+W//8].sum(); blkf=afast[r0:r0+W//8].sum()
    print("  rows %4d-%4d: %.4f"%(r0,r0+W//8,blkf/blk64))
d=np.clip(128+8*(imgs["fast"][1].astype(float)-imgs["f64"][1].astype(float)),0,255).astype(np.uint8)
rta.write_png("gpurun_out/diff_%s.png"%which, d)
