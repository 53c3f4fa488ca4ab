#!/bin/bash
# round 4: k_lz4_exec with fewer entries in flight (idle LDS per workgroup): time, L2 hit rate, fabric requests
tag=${1:-x}; out=$PWD/gpurun_out/r4_inflight_$tag; rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
args="--entries 100000 --steps 3 --warmup 1 --no-cpu"
for pad in 0 5120 15360 35840; do
  export ZPK_BENCH_LZ4_EXEC_PAD=$pad
  timeout -k 10 200 python3 $root/bench.py $args > $out/bench_$pad.json 2> $out/bench_$pad.err
  python3 - <<PY
import json
try:
    d=json.loads(open("$out/bench_$pad.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("pad $pad: %.1f GiB/s kernel_ms %.3f parse_ms %.3f exec+left_ms %.3f parity %s" % (d["value"], r["kernel_ms"], r["lz4_parse_ms"], r["kernel_ms"]-r["lz4_parse_ms"], d["parity"]["all_ranks"]))
except Exception as e: print("pad $pad unreadable", e)
PY
  timeout -k 10 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $out/p${pad}_a -o p --output-format csv -- python3 $root/bench.py $args > $out/p${pad}_a.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_64B_sum -d $out/p${pad}_b -o p --output-format csv -- python3 $root/bench.py $args > $out/p${pad}_b.log 2>&1
  mkdir -p $out/sum_$pad; cp -r $out/p${pad}_a $out/p${pad}_b $out/sum_$pad/
  python3 $root/tools/pmc_summary.py $out/sum_$pad | grep -A10 "^k_lz4_exec" | head -12
done
