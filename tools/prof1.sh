set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout -k 10 60 rocprofv3 -L > gpurun_out/prof/counters_list.txt 2>&1 || true
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -- python3 bench.py --steps 1 --warmup 0 --spp-step 4 --no-cpu-baseline > gpurun_out/prof/kt.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/prof/pmc1 -- python3 bench.py --steps 1 --warmup 0 --spp-step 4 --no-cpu-baseline > gpurun_out/prof/pmc1.log 2>&1
ls -R gpurun_out/prof | head -50
