#!/bin/bash
# round 5, step M: k_zstd_fse with eight streams per wave, second version (independent scans, computed extra bits, counts in the ring: 56 streams per CU): tests + A/B
out=gpurun_out/r05m; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_big_entries.py -m gpu -x -q -k "zstd or ZSTD or device_batch or foreign or damaged or status or big or frame" > $out/pytest.log 2>&1; rc=$?
tail -4 $out/pytest.log
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
so_of() { [ $1 = new ] && echo $PWD/zpack_amd/libzpk_codec.so || echo $PWD/zpack_amd/dev/ab_$1.so; }
for rep in 1 2; do for v in head new; do
  one $(so_of $v) ${v}_c3_$rep --workload c3_zstd_256k --steps 3 --warmup 1
done; done
for v in head new; do
  one $(so_of $v) ${v}_c3text --workload c3_zstd_256k --mix 0 --entries 30000 --steps 3 --warmup 1
  one $(so_of $v) ${v}_c4 --workload c4_mixed --steps 3 --warmup 1
done
