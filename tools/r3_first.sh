#!/bin/bash
# round 3, first GPU call: suite + baseline bench on this box, LDS micro costs, the bound diagnosis of k_lz4_wave (PMC), LDS cycles by phase
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3a/pytest_gpu.log
timeout -k 10 300 python bench.py > gpurun_out/r3a/bench_c2.json 2> gpurun_out/r3a/bench_c2.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r3a/bench_c2.json
timeout -k 10 120 tools/micro/lds_cost > gpurun_out/r3a/lds_cost.txt 2>&1; echo "lds_cost rc=$?"; cat gpurun_out/r3a/lds_cost.txt
PMC="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" tools/abl_pmc2.sh k_lz4_wave lds
for so in zpack_amd/abl_no*.so; do mv $so $so.off; done      # the second counter group on the base build only
PMC="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS" tools/abl_pmc2.sh k_lz4_wave issue
PMC="SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" tools/abl_pmc2.sh k_lz4_wave issue2
for so in zpack_amd/abl_no*.so.off; do mv $so ${so%.off}; done
tools/abl_run.sh --entries 30000 --steps 3 --warmup 1 --no-cpu --mix 0
