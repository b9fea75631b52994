# tools/phase_variants.sh "<args of phase_stats.py>" name1 name2 ...: the phase split of the counted kernel of prebuilt library variants
cd $GRAFT_REPO_ROOT
a=$1; shift
L=ray-tracing_amd/csrc/librt_amd.so
cp $L /tmp/tree.so
for v in "$@"; do
  if [ $v = tree ]; then cp /tmp/tree.so $L; else cp ray-tracing_amd/csrc/variants/$v.so $L; fi
  echo "== $v"; timeout -k 10 200 python -u tools/phase_stats.py $a 2>&1 | python3 -c "
import sys, json
d = json.load(sys.stdin)
print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k not in ('counters', 'dbg_raw', 'dbg_per_node_step', 'dbg_per_leaf_step')})
print('rays', d['counters']['rays'], 'paths', d['counters']['paths'])
"
done
cp /tmp/tree.so $L
