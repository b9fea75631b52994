# tools/lib_variants.sh "<python script and args>" name1 name2 ...: run a script with each prebuilt library variant in place of the tree's
cd $GRAFT_REPO_ROOT
a=$1; shift
L=ray-tracing_amd/csrc/librt_amd.so
cp $L /tmp/tree.so
for v in "$@"; do
  if [ $v = tree ]; then cp /tmp/tree.so $L; else cp ray-tracing_amd/csrc/variants/$v.so $L; fi
  echo "== $v"; timeout -k 10 300 python -u $a 2>&1 | tail -6
done
cp /tmp/tree.so $L
