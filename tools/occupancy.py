import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rta = importlib.import_module("ray-tracing_amd"); scenes = importlib.import_module("ray-tracing_amd.scenes")
s, cam, fl, meta = scenes.battle(W=64, tri_scale=0.05)
r = rta.Renderer(0); r.upload(s)
b, l = C.c_int(), C.c_int()
print("rc", rta.lib.rt_debug_occupancy(r._h, C.byref(b), C.byref(l)), "blocks/CU", b.value, "lds", l.value)
