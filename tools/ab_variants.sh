# A/B of kernel builds: tools/ab_variants.sh name1 name2 ... (files ray-tracing_amd/csrc/variants/<name>.so, built beforehand).
# Runs on the GPU box's scratch copy of the tree: each variant is copied over librt_amd.so and bench.py is timed.
cd $GRAFT_REPO_ROOT
L=ray-tracing_amd/csrc/librt_amd.so
cp $L /tmp/base.so
mkdir -p gpurun_out/ab
# the first run on a fresh box is ~2 % slower than the following ones: one discarded run first
timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 2 --spp-step ${SPP:-16} > gpurun_out/ab/warmup.log 2>&1 || { echo "warm-up FAILED"; tail -3 gpurun_out/ab/warmup.log; exit 1; }
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp ray-tracing_amd/csrc/variants/$v.so $L; fi
  timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 3 --spp-step ${SPP:-16} > gpurun_out/ab/$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/ab/$v.log; exit 1; }
  python - $v <<'PY'
import json,sys
l=[x for x in open("gpurun_out/ab/%s.log"%sys.argv[1]) if x.startswith("{")][-1]; d=json.loads(l)
print(sys.argv[1], d["value"], "Mpaths/s", d["roofline"]["avg_launch_ms"], "ms/launch")
PY
done
cp /tmp/base.so $L
