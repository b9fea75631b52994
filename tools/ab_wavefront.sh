# two-kernel wavefront mode under different builds: tools/ab_wavefront.sh name...
cd $GRAFT_REPO_ROOT
L=ray-tracing_amd/csrc/librt_amd.so
cp $L /tmp/base.so
mkdir -p gpurun_out/ab
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp ray-tracing_amd/csrc/variants/$v.so $L; fi
  timeout -k 10 200 python -u bench.py --no-cpu-baseline --steps 3 --mode wavefront > gpurun_out/ab/wf_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/ab/wf_$v.log; exit 1; }
  python - $v <<'PY'
import json,sys
l=[x for x in open("gpurun_out/ab/wf_%s.log"%sys.argv[1]) if x.startswith("{")][-1]; d=json.loads(l)
print("wavefront", sys.argv[1], d["value"], "Mpaths/s")
PY
done
