#!/usr/bin/env python3
"""Diagnostic: where the counted render kernel spends its wave cycles (battle scene, small spp)."""
import ctypes as C, importlib, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
rta = importlib.import_module("ray-tracing_amd"); scenes = importlib.import_module("ray-tracing_amd.scenes")
W = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
if len(sys.argv) > 3 and sys.argv[3] == "stress":  # cfg5-like: 1 M triangles + 1000 spheres
    s, cam, fl = scenes.stress(W=W, n_tris=1_000_000, n_spheres=1000)
elif len(sys.argv) > 3:  # any other scene function by name
    s, cam, fl = getattr(scenes, sys.argv[3])(W=W)[:3]
else:
    s, cam, fl, meta = scenes.battle(W=W)
r = rta.Renderer(0); r.upload(s); r.accum_clear(W, W); r.reset_counters()
rta.lib.rt_debug_count_own(r._h, 0 if os.environ.get("RT_PHASE_REFERENCE") else 1)  # the production traversal (filter tree) unless asked otherwise
r.render(cam, rta.make_params(W, W, spp, flags=fl | rta.RT_F_COUNTERS)); r.sync()
c = r.counters(); d = (C.c_uint64 * 24)(); rta.lib.rt_debug_stats(r._h, d)
dbg = [int(x) for x in d][12:20]
names = ["cyc_fresh", "cyc_roots", "cyc_node", "cyc_leaf", "cyc_finish", "cyc_shade", "node_steps", "node_lanes", "leaf_steps", "leaf_lanes", "queries", "query_lanes"]
d = dict(zip(names, [int(x) for x in d]))
tot = sum(d[k] for k in names[:6])
out = {k: round(d[k] / tot, 4) for k in names[:6]}
out.update(node_fill=d["node_lanes"] / max(1, d["node_steps"]) / 64, leaf_fill=d["leaf_lanes"] / max(1, d["leaf_steps"]) / 64,
           query_fill=d["query_lanes"] / max(1, d["queries"]) / 64, cyc_per_node_step=d["cyc_node"] / max(1, d["node_steps"]),
           cyc_per_leaf_step=d["cyc_leaf"] / max(1, d["leaf_steps"]), node_steps_per_query=d["node_steps"] / d["queries"],
           leaf_steps_per_query=d["leaf_steps"] / d["queries"], cyc_per_query_total=tot / d["queries"], ms=r.last_render_ms()[0], counters=c,
           dbg_per_node_step=[round(x / max(1, d["node_steps"]), 1) for x in dbg[:4]], dbg_per_leaf_step=[round(x / max(1, d["leaf_steps"]), 1) for x in dbg[4:]])
out['dbg_raw'] = dbg; out['cyc_total'] = tot; out['second_pass_rays'] = dbg[6]; out['second_passes'] = dbg[7]; out['queries'] = d['queries']
print(json.dumps(out, indent=1))
