#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (sums over dispatches) and stamp it with the library
build it was taken with: tools/pmc_summary2.py <dir> <config> <spp_step> <size>"""
import collections, csv, glob, hashlib, json, os, sys
d, cfg, spp_step, size = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = max(glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    disp[k].add(r.get('Dispatch_Id', r.get('Correlation_Id', '')))
kern = {}
for k, v in agg.items():
    if 'rt_' not in k:
        continue
    v = dict(v)
    v['dispatches'] = len(disp[k])
    if 'SQ_ACTIVE_INST_VALU' in v and 'SQ_THREAD_CYCLES_VALU' in v and v['SQ_ACTIVE_INST_VALU']:
        v['valu_lane_utilization'] = v['SQ_THREAD_CYCLES_VALU'] / (v['SQ_ACTIVE_INST_VALU'] * 64)
    if v.get('SQ_WAVE_CYCLES'):
        for n in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_ACTIVE_INST_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_ACTIVE_INST_ANY'):
            if n in v:
                v[n + '_frac_of_wave_cycles'] = v[n] / v['SQ_WAVE_CYCLES']
    if v.get('TCC_REQ_sum'):
        v['l2_hit_rate'] = v.get('TCC_HIT_sum', 0.0) / (v.get('TCC_HIT_sum', 0.0) + v.get('TCC_MISS_sum', 0.0) or 1.0)
    if v.get('SQ_ACTIVE_INST_LDS') and 'SQ_LDS_BANK_CONFLICT' in v:
        v['lds_bank_conflict_frac_of_lds_active'] = v['SQ_LDS_BANK_CONFLICT'] / v['SQ_ACTIVE_INST_LDS']
    kern[k] = v
h = hashlib.sha256(open(os.path.join(root, 'ray-tracing_amd', 'csrc', 'librt_amd.so'), 'rb').read()).hexdigest()[:16]
print(json.dumps({'config': cfg, 'spp_step': spp_step, 'size': size, 'lib_sha16': h,
                  'command': 'bench.py --config %s --steps 2 --warmup 1 --no-cpu-baseline --no-time-to-image' % cfg, 'kernels': kern}, indent=1))
