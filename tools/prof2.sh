set -e
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof/pmc2 gpurun_out/prof/pmc3 && mkdir -p gpurun_out/prof
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/prof/pmc2 -- python3 bench.py --steps 1 --warmup 0 --spp-step 8 --no-cpu-baseline > gpurun_out/prof/pmc2.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof/pmc3 -- python3 bench.py --steps 1 --warmup 0 --spp-step 8 --no-cpu-baseline > gpurun_out/prof/pmc3.log 2>&1
