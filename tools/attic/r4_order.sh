#!/bin/bash
# A/B on one box: decode work lists in archive order against largest entries first (ZPK_OPT_ORDER_MIN) — C4 (4 KiB ... 1 MiB, both
# methods), C2 and C3 (uniform sizes: nothing to gain, the cost of the two sort kernels).
out=gpurun_out/r04; mkdir -p $out
run() { label=$1; shift; timeout -k 10 500 python bench.py "$@" --no-cpu > $out/order_tmp.json 2> $out/order_tmp.err || { echo "$label FAILED"; tail -3 $out/order_tmp.err; return 1; }
  python3 - "$label" <<PY
import json,sys
d=json.loads(open("$out/order_tmp.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[1], round(d["value"],1), d["unit"], round(d["ms_per_step"],2), "ms/step; stages", r.get("stage_ms"), "parity", d["parity"])
PY
}
run "c4 default (1st)  " --workload c4_mixed --steps 3 --warmup 1 && \
ZPK_BENCH_ORDER_MIN=0 run "c4 archive order (ORDER_MIN=0)" --workload c4_mixed --steps 3 --warmup 1 && \
run "c4 default        " --workload c4_mixed --steps 3 --warmup 1 && \
ZPK_BENCH_ORDER_MIN=0 run "c2 ORDER_MIN=0    " --steps 20 --warmup 3 && \
run "c2 default        " --steps 20 --warmup 3 && \
ZPK_BENCH_ORDER_MIN=0 run "c3 ORDER_MIN=0    " --workload c3_zstd_256k --steps 3 --warmup 1 && \
run "c3 default        " --workload c3_zstd_256k --steps 3 --warmup 1
