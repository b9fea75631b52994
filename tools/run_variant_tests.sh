# the GPU test suite on a prebuilt library variant: tools/run_variant_tests.sh <name> [pytest args]
cd $GRAFT_REPO_ROOT
cp ray-tracing_amd/csrc/librt_amd.so /tmp/tree.so
v=$1; shift
cp ray-tracing_amd/csrc/variants/$v.so ray-tracing_amd/csrc/librt_amd.so
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread "$@" > gpurun_out/tests_$v.log 2>&1
rc=$?
cp /tmp/tree.so ray-tracing_amd/csrc/librt_amd.so
tail -5 gpurun_out/tests_$v.log
exit $rc
