cd $GRAFT_REPO_ROOT
L=ray-tracing_amd/csrc/librt_amd.so
cp $L /tmp/base.so
for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp ray-tracing_amd/csrc/variants/$v.so $L; fi
  echo $v $(timeout -k 10 200 python -u tools/stress_probe.py 1000000 1000 1024 2>&1 | grep "^{" | grep -o '"Mpaths_s": [0-9.]*, "parity_samples": [0-9]*, "parity_bad": [0-9]*, "hits_equal": [a-z]*')
done
cp /tmp/base.so $L
