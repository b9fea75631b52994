"""Diagnostic: time of each of ten consecutive 32-spp launches in a fresh process (is the first slower?)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
rta = importlib.import_module("ray-tracing_amd"); scenes = importlib.import_module("ray-tracing_amd.scenes")
W = 1024
s, cam, fl, meta = scenes.battle(W=W)
r = rta.Renderer(0); r.upload(s)
acc = torch.zeros((W, W, 3), dtype=torch.float64, device="cuda")
out = []
for k in range(10):
    r.render(cam, rta.make_params(W, W, 32, spp_offset=32 * k, flags=fl), acc.data_ptr()); r.sync()
    out.append(round(r.last_render_ms()[0], 2))
print(out)
r.close()
