#!/bin/bash
# round 5, step G: LZ4 entries that are mostly runs routed to k_lz4_left by k_classify (k_lz4_wave = the round-4 code, bit for bit): GPU suite, then A/B on one box
out=gpurun_out/r05g; mkdir -p $out
timeout -k 10 1100 python -m pytest tests/test_gpu_codec.py tests/test_gpu_contexts.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
one() {  # so label args...
  so=$1; label=$2; shift 2
  ZPACK_AMD_CODEC_SO=$so timeout -k 10 400 python bench.py "$@" --no-cpu > $out/$label.json 2> $out/$label.err || { tail -5 $out/$label.err; return 1; }
  python3 - <<PY
import json
d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$label: %.1f %s  %.3f ms/step  kernel %.3f ms %s runs-entries %s parity %s" % (d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"], r.get("stage_ms"), r.get("lz4_entries_of_long_runs"), d["parity"]["all_ranks"]))
PY
}
so_of() { [ $1 = new ] && echo $PWD/zpack_amd/libzpk_codec.so || echo $PWD/zpack_amd/dev/ab_$1.so; }
for rep in 1 2; do for v in head new; do
  one $(so_of $v) ${v}_mix_$rep --steps 10 --warmup 3
  one $(so_of $v) ${v}_text_$rep --mix 0 --steps 8 --warmup 2
done; done
for v in head new left8; do
  one $(so_of $v) ${v}_runs --mix 3 --steps 8 --warmup 2
done
for v in head new; do
  one $(so_of $v) ${v}_records --mix 1 --steps 8 --warmup 2
  one $(so_of $v) ${v}_c4 --workload c4_mixed --steps 3 --warmup 1
done
one $(so_of left8) left8_mix --steps 10 --warmup 3
one $(so_of left8) left8_c4 --workload c4_mixed --steps 3 --warmup 1
