#!/bin/bash
# A/B on one box: within a size class, entries that did not compress go last (ZPK_OPT_ORDER_FAST_LAST) — C2 twice over, C3, C4.
out=gpurun_out/r04; mkdir -p $out
run() { label=$1; shift; timeout -k 10 500 python bench.py "$@" --no-cpu > $out/fl_tmp.json 2> $out/fl_tmp.err || { echo "$label FAILED"; tail -3 $out/fl_tmp.err; return 1; }
  python3 - "$label" <<PY
import json,sys
d=json.loads(open("$out/fl_tmp.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], round(d["value"],1), d["unit"], round(d["ms_per_step"],3), "ms/step; stages", [round(x,2) for x in r.get("stage_ms")], "parity", all(v in (True, None) for v in d["parity"].values()))
PY
}
for i in 1 2; do
ZPK_BENCH_ORDER_FAST=0 run "c2 fast_last=0" --steps 20 --warmup 3 && ZPK_BENCH_ORDER_FAST=1 run "c2 fast_last=1" --steps 20 --warmup 3 || exit 1
done
ZPK_BENCH_ORDER_FAST=0 run "c3 fast_last=0" --workload c3_zstd_256k --steps 3 --warmup 1 && \
ZPK_BENCH_ORDER_FAST=1 run "c3 fast_last=1" --workload c3_zstd_256k --steps 3 --warmup 1 && \
ZPK_BENCH_ORDER_FAST=0 run "c4 fast_last=0" --workload c4_mixed --steps 3 --warmup 1 && \
ZPK_BENCH_ORDER_FAST=1 run "c4 fast_last=1" --workload c4_mixed --steps 3 --warmup 1
