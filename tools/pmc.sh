#!/bin/bash
# Hardware counters for the decode kernels of one bench.py workload, one rocprofv3 --pmc pass per counter
# group (never combined with trace domains).  Usage on the GPU box:  tools/pmc.sh <outdir> [bench args...]
# Summarise with tools/pmc_summary.py <outdir>.
out=${1:-gpurun_out/pmc}; shift
args=${@:---entries 20000 --steps 2 --warmup 1 --no-cpu}
root=$PWD
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $grp -d "$root/$out/g$i" -o p --output-format csv -- python3 "$root/bench.py" $args > "$root/$out/g$i.log" 2>&1
    echo "group $i ($grp): rc=$?"
done
