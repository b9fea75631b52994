set -e
cd $GRAFT_REPO_ROOT/ray-tracing_amd/csrc
cp librt_amd.so /tmp/good.so
for E in 1 3; do cp librt_amd_exp$E.so librt_amd.so; (cd ../..; timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --spp-step 16 --no-cpu-baseline 2>&1 | tail -1 | cut -c60-130); done
cp /tmp/good.so librt_amd.so
