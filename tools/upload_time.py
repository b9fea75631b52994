#!/usr/bin/env python3
"""Set-up cost of a configuration: scene construction, rt_upload_scene (stage by stage with RT_UPLOAD_TIMING=1), first render."""
import importlib, os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rta = importlib.import_module("ray-tracing_amd"); scenes = importlib.import_module("ray-tracing_amd.scenes")
which = sys.argv[1] if len(sys.argv) > 1 else "battle"
t0 = time.time()
out = scenes.stress() if which == "stress" else scenes.battle(W=1024)
s, cam = out[0], out[1]
t1 = time.time()
r = rta.Renderer(0)
t2 = time.time()
r.upload(s)
t3 = time.time()
r.upload(s)
t4 = time.time()
print(json.dumps(dict(scene=which, build_s=round(t1 - t0, 3), create_s=round(t2 - t1, 3), upload_s=round(t3 - t2, 3), upload_again_s=round(t4 - t3, 3))))
