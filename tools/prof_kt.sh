set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof/kt && mkdir -p gpurun_out/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python3 bench.py --steps 1 --warmup 0 --spp-step ${SPP:-16} --no-cpu-baseline > gpurun_out/prof/kt.log 2>&1
tail -1 gpurun_out/prof/kt.log | cut -c1-200
head -8 gpurun_out/prof/kt/*/*_kernel_stats.csv
