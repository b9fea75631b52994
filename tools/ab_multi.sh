# A/B of prebuilt library variants over several bench configurations: tools/ab_multi.sh "cfg2 cfg3 cfg5" name1 name2 ...
# ("tree" = the library in the tree; others: ray-tracing_amd/csrc/variants/<name>.so). One discarded warm-up run first.
cd $GRAFT_REPO_ROOT
cfgs=$1; shift
L=ray-tracing_amd/csrc/librt_amd.so
cp $L /tmp/tree.so
mkdir -p gpurun_out/ab
timeout -k 10 300 python -u bench.py --config cfg4 --no-cpu-baseline --no-time-to-image --steps 2 > gpurun_out/ab/warmup.log 2>&1 || { echo "warm-up FAILED"; tail -3 gpurun_out/ab/warmup.log; exit 1; }
for cfg in $cfgs; do
for v in "$@"; do
  if [ $v = tree ]; then cp /tmp/tree.so $L; else cp ray-tracing_amd/csrc/variants/$v.so $L; fi
  timeout -k 10 300 python -u bench.py --config $cfg --no-cpu-baseline --no-time-to-image --steps 3 > gpurun_out/ab/${cfg}_$v.log 2>&1 || { echo "$cfg $v FAILED"; tail -3 gpurun_out/ab/${cfg}_$v.log; cp /tmp/tree.so $L; exit 1; }
  python - $cfg $v <<'PY'
import json,sys
l=[x for x in open("gpurun_out/ab/%s_%s.log"%(sys.argv[1],sys.argv[2])) if x.startswith("{")][-1]; d=json.loads(l)
print(sys.argv[1], sys.argv[2], d["value"], "Mpaths/s", d["roofline"]["avg_launch_ms"], "ms/launch")
PY
done
done
cp /tmp/tree.so $L
