# tools/fast_variants.sh "<args of fast_try.py>" name1 name2 ...: the single-precision kernel of prebuilt library variants against the double-precision render
cd $GRAFT_REPO_ROOT
a=$1; shift
L=ray-tracing_amd/csrc/librt_amd.so
cp $L /tmp/tree.so
mkdir -p gpurun_out
for v in "$@"; do
  if [ $v = tree ]; then cp /tmp/tree.so $L; else cp ray-tracing_amd/csrc/variants/$v.so $L; fi
  echo "== $v"; timeout -k 10 200 python -u tools/fast_try.py $a 2>&1 | grep -E "Mpaths|image diff" 
done
cp /tmp/tree.so $L
