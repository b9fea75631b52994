# set-up timing of a prebuilt library variant: tools/upload_variant.sh <name> <battle|stress>
cd $GRAFT_REPO_ROOT
cp ray-tracing_amd/csrc/variants/$1.so ray-tracing_amd/csrc/librt_amd.so
RT_UPLOAD_TIMING=1 timeout -k 10 200 python tools/upload_time.py $2 > gpurun_out/up_$1_$2.log 2>&1
tail -16 gpurun_out/up_$1_$2.log
