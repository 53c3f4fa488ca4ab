#!/bin/bash
# developer: phase cycles of k_lz4_wave (ZPK_STATS builds under zpack_amd/dev) + instruction / LDS counters of every abl_*.so
mkdir -p gpurun_out/r3c
for mix in 0 1; do
  ZPACK_AMD_CODEC_SO=$PWD/zpack_amd/dev/st.so timeout -k 10 200 python3 tools/lw_stats.py 20000 $mix 2>&1 | tail -8
  LW_PARSE=1 ZPACK_AMD_CODEC_SO=$PWD/zpack_amd/dev/stp.so timeout -k 10 200 python3 tools/lw_stats.py 20000 $mix 2>&1 | tail -3
done | tee gpurun_out/r3c/lw_stats.txt
PMC="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" tools/abl_pmc2.sh k_lz4_wave r3c
PMC="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM" tools/abl_pmc2.sh k_lz4_wave r3c2
