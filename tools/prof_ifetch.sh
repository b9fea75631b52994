# instruction-fetch / scalar-cache counter pass for the library in the tree
set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pq_ifetch
rm -rf $O && mkdir -p $O
B="python3 bench.py --steps 2 --warmup 1 --spp-step 16 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $O/a -- $B > $O/a.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_BUSY_CYCLES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQC_TC_STALL SQC_TC_INST_REQ --output-format csv -d $O/b -- $B > $O/b.log 2>&1
for x in a b; do python3 tools/pmc_summary.py $O/$x > $O/$x.json; python3 - $O/$x.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); k=[x for x in d if "rt_render_kernel<false, true>" in x][0]; print(json.dumps(d[k]))
PY
done
