#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (sums over dispatches)."""
import collections, csv, glob, json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof/pmc1'
f = max(glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    disp[k].add(r.get('Dispatch_Id', r.get('Correlation_Id', '')))
out = {}
for k, v in agg.items():
    if 'rt_' not in k:
        continue
    v = dict(v)
    v['dispatches'] = len(disp[k])
    if 'SQ_ACTIVE_INST_VALU' in v and 'SQ_THREAD_CYCLES_VALU' in v:
        v['valu_lane_utilization'] = v['SQ_THREAD_CYCLES_VALU'] / (v['SQ_ACTIVE_INST_VALU'] * 64)
    if 'SQ_WAVE_CYCLES' in v:
        for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_ANY'):
            if n in v:
                v[n + '_frac_of_wave_cycles'] = v[n] / v['SQ_WAVE_CYCLES']
    out[k] = v
print(json.dumps(out, indent=1))
