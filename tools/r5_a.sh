#!/bin/bash
# round 5, step A: grouped cooperative copies (seq_exec.h coop_copy_rows): LZ4 tests + the stream tests, then C2 per class
out=gpurun_out/r05a; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_codec.py tests/test_gpu_zpack_api.py -m gpu -x -q -k "lz4 or LZ4 or window_sizes or device_batch or damaged or foreign or shapes" > $out/pytest.log 2>&1; rc=$?
tail -5 $out/pytest.log
[ $rc -eq 0 ] || exit $rc
for mix in 3 0 1 -1; do
  timeout -k 10 300 python bench.py --mix $mix --steps 8 --warmup 2 --no-cpu > $out/c2_mix$mix.json 2> $out/c2_mix$mix.err || { tail -5 $out/c2_mix$mix.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$out/c2_mix$mix.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("mix $mix: %.1f GiB/s  %.3f ms/step  kernel %.3f ms  parity %s" % (d["value"], d["ms_per_step"], r["kernel_ms"], d["parity"]["all_ranks"]))
PY
done
