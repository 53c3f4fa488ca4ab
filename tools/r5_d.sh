#!/bin/bash
# round 5, step D: the whole GPU suite on the lean k_lz4_wave + k_lz4_left split, then A/B against the round-4 kernels (zpack_amd/dev/ab_head.so) on one box
out=gpurun_out/r05d; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
one() {  # so label args...
  so=$1; label=$2; shift 2
  ZPACK_AMD_CODEC_SO=$so timeout -k 10 400 python bench.py "$@" --no-cpu > $out/$label.json 2> $out/$label.err || { tail -5 $out/$label.err; return 1; }
  python3 - <<PY
import json
d=json.loads(open("$out/$label.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$label: %.1f %s  %.3f ms/step  kernel %.3f ms %s parity %s" % (d["value"], d["unit"], d["ms_per_step"], r["kernel_ms"], r.get("stage_ms"), d["parity"]["all_ranks"]))
PY
}
H=$PWD/zpack_amd/dev/ab_head.so; N=$PWD/zpack_amd/libzpk_codec.so
for rep in 1 2; do
  one $H head_c2_$rep --steps 8 --warmup 2
  one $N new_c2_$rep --steps 8 --warmup 2
  one $H head_text_$rep --mix 0 --steps 8 --warmup 2
  one $N new_text_$rep --mix 0 --steps 8 --warmup 2
done
one $H head_runs --mix 3 --steps 8 --warmup 2
one $N new_runs --mix 3 --steps 8 --warmup 2
one $H head_records --mix 1 --steps 8 --warmup 2
one $N new_records --mix 1 --steps 8 --warmup 2
one $H head_c4 --workload c4_mixed --steps 3 --warmup 1
one $N new_c4 --workload c4_mixed --steps 3 --warmup 1
