#!/bin/bash
# round 5, step C: A/B of the cooperative-copy variants (seq_exec.h SEQ_COOP_V = 0 round-4 code / 1 grouped matches / 2 + grouped long literals) on one box
out=gpurun_out/r05c; mkdir -p $out
one() {  # so label args...
  so=$1; label=$2; shift 2
  ZPACK_AMD_CODEC_SO=$so timeout -k 10 400 python bench.py "$@" --no-cpu > $out/$label.json 2> $out/$label.err || { tail -5 $out/$label.err; return 1; }
  python3 - <<PY
import json
d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$label: %.1f %s  %.3f ms/step  kernel %.3f ms %s parity %s" % (d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"], r.get("stage_ms"), d["parity"]["all_ranks"]))
PY
}
for rep in 1 2; do for v in 0 1 2; do
  one $PWD/zpack_amd/dev/ab_v$v.so v${v}_text_$rep --mix 0 --steps 8 --warmup 2
done; done
for v in 0 1 2; do
  one $PWD/zpack_amd/dev/ab_v$v.so v${v}_runs --mix 3 --steps 8 --warmup 2
  one $PWD/zpack_amd/dev/ab_v$v.so v${v}_mix --steps 8 --warmup 2
done
