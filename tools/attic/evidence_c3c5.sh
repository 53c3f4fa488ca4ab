#!/bin/bash
# The C3 and C5 parts of the evidence alone: PMC passes of C3 + the bench line quoting them, then tools/pmc_c5.sh.
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
root=$PWD
(cd /tmp && export TMPDIR=/tmp && for grp in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do g=$(echo $grp | cut -d' ' -f1); rm -rf $root/$out/pmc_c3/$g; timeout -k 10 600 rocprofv3 --pmc $grp -d $root/$out/pmc_c3/$g -o p --output-format csv -- python3 $root/bench.py --workload c3_zstd_256k --steps 2 --warmup 1 --no-cpu > $root/$out/pmc_c3_$g.log 2>&1; echo "pmc c3 $g rc=$?"; done)
python tools/pmc_summary.py $out/pmc_c3 --json $out/pmc_c3_zstd_256k.json --entries 100000 --workload c3_zstd_256k > $out/pmc_c3_zstd_256k.txt
cp $out/pmc_c3_zstd_256k.json $out/pmc_c3_zstd_256k.txt profiles/$tag/
timeout -k 10 600 python bench.py --workload c3_zstd_256k --steps 3 --warmup 1 > $out/${tag}_c3_zstd_bench.json 2> $out/c3.err; echo "bench c3 rc=$?"
rm -rf $out/pmc_c5
tools/pmc_c5.sh $tag | tail -c 600
