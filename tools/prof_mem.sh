set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof/mem1 gpurun_out/prof/mem2 && mkdir -p gpurun_out/prof
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY --output-format csv -d gpurun_out/prof/mem1 -- python3 bench.py --steps 1 --warmup 0 --spp-step 8 --no-cpu-baseline > gpurun_out/prof/mem1.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_DRAM_sum --output-format csv -d gpurun_out/prof/mem2 -- python3 bench.py --steps 1 --warmup 0 --spp-step 8 --no-cpu-baseline > gpurun_out/prof/mem2.log 2>&1
