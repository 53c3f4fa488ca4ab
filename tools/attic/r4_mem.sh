#!/bin/bash
# round 4: memory-side counters of the LZ4 kernels (TA / TCP / TCC), two-stage path and one-kernel path (ZPK_BENCH_LZ4_TWO=never)
tag=${1:-x}; out=$PWD/gpurun_out/r4_mem_$tag; rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters_list.txt 2>&1
args="--entries 100000 --steps 2 --warmup 1 --no-cpu"
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_WRREQ_sum TCC_BUSY_avr TCC_TAG_STALL_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  for mode in two one; do
    if [ $mode = one ]; then export ZPK_BENCH_LZ4_TWO=never; else unset ZPK_BENCH_LZ4_TWO; fi
    timeout -k 10 200 rocprofv3 --pmc $grp -d $out/${mode}_g$i -o p --output-format csv -- python3 $root/bench.py $args > $out/${mode}_g$i.log 2>&1
    echo "$mode group $i rc=$?"
  done
done
for mode in two one; do
  mkdir -p $out/sum_$mode; for d in $out/${mode}_g*; do [ -d $d ] && cp -r $d $out/sum_$mode/; done
  echo "=== $mode"; python3 $root/tools/pmc_summary.py $out/sum_$mode | grep -A30 "^k_lz4_exec\|^k_lz4_wave\|^k_lz4_parse" | grep -v "^k_lz4_retry" | head -100
done
