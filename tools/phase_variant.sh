# phase statistics of a prebuilt kernel variant: tools/phase_variant.sh <name> [W spp [stress]]
cd $GRAFT_REPO_ROOT
cp ray-tracing_amd/csrc/variants/$1.so ray-tracing_amd/csrc/librt_amd.so
timeout -k 10 120 python -u tools/phase_stats.py ${2:-1024} ${3:-8} $4 > gpurun_out/phase_$1.json 2>&1
head -20 gpurun_out/phase_$1.json; tail -4 gpurun_out/phase_$1.json
