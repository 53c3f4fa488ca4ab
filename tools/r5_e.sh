#!/bin/bash
# round 5, step E: compiler-flag lottery on the decode kernels (k_lz4_wave's code for the common path moved 6 % with unrelated code around it): one box, text + mix + c3
out=gpurun_out/r05e; mkdir -p $out
one() {  # so label args...
  so=$1; label=$2; shift 2
  ZPACK_AMD_CODEC_SO=$so timeout -k 10 400 python bench.py "$@" --no-cpu > $out/$label.json 2> $out/$label.err || { tail -5 $out/$label.err; return 1; }
  python3 - <<PY
import json
d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$label: %.1f %s  %.3f ms/step  kernel %.3f ms %s parity %s" % (d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"], r.get("stage_ms"), d["parity"]["all_ranks"]))
PY
}
for v in "$@"; do
  so=$PWD/zpack_amd/dev/ab_$v.so; [ $v = new ] && so=$PWD/zpack_amd/libzpk_codec.so
  one $so ${v}_text --mix 0 --steps 6 --warmup 2
  one $so ${v}_mix --steps 6 --warmup 2
  one $so ${v}_c3 --workload c3_zstd_256k --entries 30000 --steps 2 --warmup 1
done
