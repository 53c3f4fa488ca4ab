#!/bin/bash
# The C2 (headline) part of tools/round2_evidence.sh alone: GPU tests, kernel trace, PMC passes, then the default bench line quoting them.
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
root=$PWD
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $out/${tag}_pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $out/${tag}_pytest_gpu.log
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $root/$out/trace_c2_lz4 -o t --output-format csv -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu > $root/$out/trace_c2_lz4_bench.json 2> $root/$out/trace_c2_lz4.err; echo "trace rc=$?"; f=$(find $root/$out/trace_c2_lz4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $root/$out/${tag}_c2_lz4_kernel_stats.csv)
(cd /tmp && export TMPDIR=/tmp && for grp in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do g=$(echo $grp | cut -d' ' -f1); rm -rf $root/$out/pmc_c2/$g; timeout -k 10 600 rocprofv3 --pmc $grp -d $root/$out/pmc_c2/$g -o p --output-format csv -- python3 $root/bench.py --steps 2 --warmup 1 --no-cpu > $root/$out/pmc_c2_$g.log 2>&1; echo "pmc c2 $g rc=$?"; done)
python tools/pmc_summary.py $out/pmc_c2 --json $out/pmc_c2_lz4_64k.json --entries 100000 --workload c2_lz4_64k > $out/pmc_c2_lz4_64k.txt
cp $out/pmc_c2_lz4_64k.json $out/pmc_c2_lz4_64k.txt profiles/$tag/
timeout -k 10 600 python bench.py > $out/${tag}_c2_lz4_bench.json 2> $out/c2.err; echo "bench c2 rc=$?"
tail -c 1200 $out/${tag}_c2_lz4_bench.json
