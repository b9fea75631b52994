# counter pass over tools/fast_try.py (both kernels on the same scene): tools/prof_fast.sh <W> <scene> <spp>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_fast
rm -rf $O && mkdir -p $O
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc1 -- python3 tools/fast_try.py $1 $2 $3 > $O/pmc1.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc2 -- python3 tools/fast_try.py $1 $2 $3 > $O/pmc2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
for f in ("pmc1","pmc2"):
    fn=max(glob.glob("gpurun_out/prof_fast/%s/**/*_counter_collection.csv"%f, recursive=True), key=os.path.getmtime)
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(set)
    for r in csv.DictReader(open(fn)):
        k=r["Kernel_Name"].split("(")[0]
        if "rt_render" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        wc=v.get("SQ_WAVE_CYCLES",1)
        print(f, k[:40], "dispatches", len(n[k]), {a: round(b/wc,4) for a,b in v.items() if a!="SQ_WAVE_CYCLES"}, "wave_cycles %.3e"%wc, "insts_valu %.3e"%v.get("SQ_INSTS_VALU",0))
PY
