#!/bin/bash
# round 4: A/B of the stage-2 executors of the two-stage LZ4 path (output slot in memory + running hash / LDS window) and the one-kernel path
tag=${1:-x}; out=$PWD/gpurun_out/r4_ab_$tag; rm -rf $out; mkdir -p $out
root=$PWD
if [ -z "$2" ]; then
  timeout -k 10 420 python -m pytest tests/test_gpu_lz4_two_stage.py -x -q -m gpu > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log
  tail -3 $out/tests.log
fi
cd /tmp && export TMPDIR=/tmp
args="--entries 100000 --steps 5 --warmup 2 --no-cpu"
for mode in one g w; do
  unset ZPK_BENCH_LZ4_TWO ZPK_BENCH_LZ4_EXEC_WINDOW
  [ $mode = one ] && export ZPK_BENCH_LZ4_TWO=never
  [ $mode = w ] && export ZPK_BENCH_LZ4_EXEC_WINDOW=1 ZPK_BENCH_LZ4_TWO=always
  [ $mode = g ] && export ZPK_BENCH_LZ4_EXEC_WINDOW=0 ZPK_BENCH_LZ4_TWO=always
  timeout -k 10 200 python3 $root/bench.py $args > $out/bench_$mode.json 2> $out/bench_$mode.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$out/bench_$mode.json").read().strip().splitlines()[-1]); r=d["roofline"]; p=r.get("lz4_parse_ms") or 0.0
    print("$mode: %.1f GiB/s kernel_ms %.3f parse_ms %.3f rest_ms %.3f parity %s" % (d["value"], r["kernel_ms"], p, r["kernel_ms"]-p, d["parity"]["all_ranks"]))
except Exception as e: print("$mode unreadable", e)
PY
  if [ $mode != one ]; then
    timeout -k 10 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $out/${mode}_a -o p --output-format csv -- python3 $root/bench.py $args > $out/${mode}_a.log 2>&1
    timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL TCC_EA0_WRREQ_sum -d $out/${mode}_b -o p --output-format csv -- python3 $root/bench.py $args > $out/${mode}_b.log 2>&1
    mkdir -p $out/sum_$mode; cp -r $out/${mode}_a $out/${mode}_b $out/sum_$mode/
    python3 $root/tools/pmc_summary.py $out/sum_$mode | grep -A14 "^k_lz4_exec" | head -16
  fi
done
