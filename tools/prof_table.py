#!/usr/bin/env python3
"""One line per configuration from the committed profile summaries profiles/r3_<cfg>_*."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for cfg in ("cfg1", "cfg2", "cfg3", "cfg4", "cfg5"):
    try:
        b = json.loads(open(os.path.join(root, "profiles", "r3_%s_bench.json" % cfg)).read().strip().split("\n")[-1])
    except OSError:
        continue
    kn = b["roofline"]["kernel"].split("<")[0]
    row = {"Mpaths/s": b["value"], "ms/launch": b["roofline"]["avg_launch_ms"], "frac": b["roofline"]["frac"], "frac_fp64": b["roofline"]["frac_fp64_part"], "frac_fp32": b["roofline"]["frac_fp32_part"],
           "rays/path": b["roofline"]["rays_per_path"], "TFLOPs": b["roofline"]["achieved"], "ref_equiv_frac": b["roofline"]["reference_semantics"]["frac_equivalent"]}
    for f in ("pmc1", "pmc2", "mem1", "mem2", "mem3"):
        d = json.load(open(os.path.join(root, "profiles", "r3_%s_%s.json" % (cfg, f))))["kernels"]
        k = max((x for x in d if kn in x and "true" not in x.split("<")[-1].split(",")[0]), key=lambda x: d[x].get("dispatches", 0))
        v = d[k]
        if f == "pmc1":
            row.update(valu_active=round(v["SQ_ACTIVE_INST_VALU_frac_of_wave_cycles"], 3), wait_any=round(v["SQ_WAIT_ANY_frac_of_wave_cycles"], 3), lane_util=round(v["valu_lane_utilization"], 3))
        if f == "pmc2":
            row.update(issue_any=round(v["SQ_ACTIVE_INST_ANY_frac_of_wave_cycles"], 3), issue_scalar=round(v["SQ_ACTIVE_INST_SCA_frac_of_wave_cycles"], 3),
                       issue_lds=round(v["SQ_ACTIVE_INST_LDS_frac_of_wave_cycles"], 3))
        if f == "mem1":
            row.update(l2_hit=round(v["l2_hit_rate"], 3), lds_conflict_of_lds_active=round(v["lds_bank_conflict_frac_of_lds_active"], 3),
                       lds_active_of_wave_cycles=None)
        if f == "mem2":
            row.update(fetch_GB_per_launch=round(v["FETCH_SIZE"] * 1024 * 1.39 / v["dispatches"] / 1e9, 3))  # x1.39: calibrated for 128-byte gathers
        if f == "mem3":
            row.update(write_MB_per_launch=round(v["WRITE_SIZE"] * 1024 / v["dispatches"] / 1e6, 1))
    row["HBM_side_GBs"] = round(row["fetch_GB_per_launch"] / (row["ms/launch"] / 1e3), 1)
    row.pop("lds_active_of_wave_cycles")
    print(cfg, json.dumps(row))
