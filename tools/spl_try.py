"""Time of a 256-sample render of the battle scene (one call), for libraries built with different samples per launch."""
import importlib, sys, numpy as np
sys.path.insert(0, "/root/repo")
rta = importlib.import_module("ray-tracing_amd"); scenes = importlib.import_module("ray-tracing_amd.scenes")
W, spp = 1024, 256
s, cam, flags = scenes.battle(W=W)[:3]
r = rta.Renderer(0); r.upload(s)
for it in range(3):
    r.accum_clear(W, W); r.render(cam, rta.make_params(W, W, spp, flags=flags)); r.sync()
    ms, nl = r.last_render_ms()
    print("battle %d spp in %d launch(es): %.2f ms, %.1f Mpaths/s" % (spp, nl, ms, W * W * spp / ms / 1e3), flush=True)
a = r.accum_download(W, W)
import hashlib; print("accumulator sha", hashlib.sha256(a.tobytes()).hexdigest()[:16])
