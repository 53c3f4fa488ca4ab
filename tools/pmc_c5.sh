#!/bin/bash
# PMC passes of the C5 write-path workload (same groups as tools/round2_evidence.sh) + the bench line that quotes them.
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
root=$PWD
(cd /tmp && export TMPDIR=/tmp && for grp in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do g=$(echo $grp | cut -d' ' -f1); timeout -k 10 600 rocprofv3 --pmc $grp -d $root/$out/pmc_c5/$g -o p --output-format csv -- python3 $root/bench.py --workload c5_zstd1_1m --steps 2 --warmup 1 --no-cpu > $root/$out/pmc_c5_$g.log 2>&1; echo "pmc c5 $g rc=$?"; done)
python tools/pmc_summary.py $out/pmc_c5 --json $out/pmc_c5_zstd1_1m.json --entries 12500 --workload c5_zstd1_1m > $out/pmc_c5_zstd1_1m.txt
cp $out/pmc_c5_zstd1_1m.json $out/pmc_c5_zstd1_1m.txt profiles/$tag/
timeout -k 10 1100 python bench.py --workload c5_zstd1_1m --steps 3 --warmup 1 > $out/${tag}_c5_zstd1_bench.json 2> $out/c5.err; echo "bench c5 rc=$?"
tail -c 1500 $out/${tag}_c5_zstd1_bench.json
