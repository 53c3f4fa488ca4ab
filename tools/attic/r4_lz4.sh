#!/bin/bash
# round 4: the two-stage LZ4 path — its tests, the C2 bench line, a kernel trace and one instruction-counter pass.
# usage (GPU box): tools/r4_lz4.sh <tag> [skip-tests]
tag=${1:-x}; out=$PWD/gpurun_out/r4_$tag; rm -rf $out; mkdir -p $out
root=$PWD
if [ -z "$2" ]; then
  timeout -k 10 420 python -m pytest tests/test_gpu_lz4_two_stage.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log
  tail -4 $out/tests.log
fi
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu > $out/bench.json 2> $out/bench.err || tail -5 $out/bench.err
python3 - <<PY
import json
try:
    d=json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    r=d["roofline"]; print("C2 value %.1f GiB/s  ms/step %.3f  kernel_ms %.3f parse_ms %s frac %.4f parity %s stats %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r.get("lz4_parse_ms"), r["frac"], d["parity"]["all_ranks"], {k:v for k,v in d["decode_stats"].items() if k.startswith("lz4")}))
except Exception as e: print("bench line unreadable:", e)
PY
cd /tmp && export TMPDIR=/tmp
args="--entries 100000 --steps 3 --warmup 1 --no-cpu"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $root/bench.py $args > $out/kt.log 2>&1
echo "kernel-trace rc=$?"
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-8 "$f" | head -8
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out/pmc1 -o p --output-format csv -- python3 $root/bench.py $args > $out/pmc1.log 2>&1
echo "pmc1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out/pmc2 -o p --output-format csv -- python3 $root/bench.py $args > $out/pmc2.log 2>&1
echo "pmc2 rc=$?"
timeout -k 10 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $out/pmc3 -o p --output-format csv -- python3 $root/bench.py $args > $out/pmc3.log 2>&1
echo "pmc3 rc=$?"
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT -d $out/pmc4 -o p --output-format csv -- python3 $root/bench.py $args > $out/pmc4.log 2>&1
echo "pmc4 rc=$?"
python3 $root/tools/pmc_summary.py $out > $out/pmc_summary.txt 2>&1; grep -A30 "^k_lz4_exec\|^k_lz4_parse" $out/pmc_summary.txt | grep -v "^k_lz4_left" | head -70
