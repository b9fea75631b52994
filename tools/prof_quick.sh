# quick counter passes for a prebuilt kernel variant ("base" = the library in the tree): tools/prof_quick.sh <name>
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
v=$1
if [ "$v" != base ]; then cp ray-tracing_amd/csrc/variants/$v.so ray-tracing_amd/csrc/librt_amd.so; fi
O=gpurun_out/pq_$v
rm -rf $O && mkdir -p $O
B="python3 bench.py --steps 2 --warmup 1 --spp-step 16 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc1 -- $B > $O/pmc1.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE GRBM_TA_BUSY TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $O/pmc2 -- $B > $O/pmc2.log 2>&1
python3 tools/pmc_summary.py $O/pmc1 > $O/pmc1.json
python3 tools/pmc_summary.py $O/pmc2 > $O/pmc2.json
python3 - $O <<'PY'
import json,sys
for f in ("pmc1","pmc2"):
    d=json.load(open(sys.argv[1]+"/"+f+".json"))
    k=[x for x in d if "rt_render_kernel<false, true>" in x][0]
    print(f, json.dumps(d[k]))
PY
