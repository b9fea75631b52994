# rocprofv3 evidence for one bench configuration: kernel trace + stats, then counter passes (a run of its own each).
#   tools/prof_config.sh cfg4 64 1024      -> gpurun_out/prof_cfg4/{kernel_stats.csv, pmc1.json, pmc2.json, mem1.json, mem2.json, mem3.json, bench.json}
set -e
cfg=$1; spp=$2; size=$3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_$cfg
rm -rf $O && mkdir -p $O
B="python3 bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-time-to-image"
timeout -k 10 300 $B > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/kt.log 2>&1
cp $(find $O/kt -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/pmc1 -- $B > $O/pmc1.log 2>&1
python3 tools/pmc_summary2.py $O/pmc1 $cfg $spp $size > $O/pmc1.json
# where a wave's cycles go by instruction type (issue-slot accounting)
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc2 -- $B > $O/pmc2.log 2>&1
python3 tools/pmc_summary2.py $O/pmc2 $cfg $spp $size > $O/pmc2.json
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS --output-format csv -d $O/mem1 -- $B > $O/mem1.log 2>&1
python3 tools/pmc_summary2.py $O/mem1 $cfg $spp $size > $O/mem1.json
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/mem2 -- $B > $O/mem2.log 2>&1
python3 tools/pmc_summary2.py $O/mem2 $cfg $spp $size > $O/mem2.json
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/mem3 -- $B > $O/mem3.log 2>&1
python3 tools/pmc_summary2.py $O/mem3 $cfg $spp $size > $O/mem3.json
rm -rf $O/kt $O/pmc1 $O/pmc2 $O/mem1 $O/mem2 $O/mem3
head -c 600 $O/kernel_stats.csv
