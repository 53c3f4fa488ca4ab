#!/bin/bash
# round 5: per-class rates of the Zstandard decode (c3) and encode (c5) workloads — looking for a class that is out of line (as LZ4 `runs` was)
out=gpurun_out/r05k; mkdir -p $out
for mix in 0 1 2 3; do
  timeout -k 10 300 python bench.py --workload c3_zstd_256k --mix $mix --entries 25000 --steps 3 --warmup 1 --no-cpu > $out/c3_m$mix.json 2> $out/c3_m$mix.err || { tail -3 $out/c3_m$mix.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("$out/c3_m$mix.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("c3 mix $mix: %.1f %s  %.3f ms/step  stages %s ratio %.3f parity %s" % (d["value"], d["unit"], d["ms_per_step"], r.get("stage_ms"), d["config"].get("comp_ratio",0), d["parity"]["all_ranks"]))
PY
done
for mix in 0 1 2 3; do
  timeout -k 10 300 python bench.py --workload c5_zstd1_1m --mix $mix --entries 3000 --steps 2 --warmup 1 --no-cpu > $out/c5_m$mix.json 2> $out/c5_m$mix.err || { tail -3 $out/c5_m$mix.err; continue; }
  python3 - <<PY
import json
d=json.loads(open("$out/c5_m$mix.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("c5 mix $mix: %.1f %s  %.3f ms/step  stages %s ratio %s parity %s" % (d["value"], d["unit"], d["ms_per_step"], r.get("stage_ms"), d["config"].get("comp_ratio"), d["parity"]["all_ranks"]))
PY
done
