#!/bin/bash
# round 5: the encoder with 8-bit tags beside its hash positions against the product build (c5_zstd1_1m, A/B/A/B).  The variant is built OUTSIDE
# the tree (the sources must keep their hash): copy zpack_amd/csrc + include to a temporary directory, add `u8 tag[1 << HLOG]` to Lz4EncSharedT,
# keep (product >> (24 - HLOG)) & 0xFF of a position's hash there, skip the probe of a candidate whose tag differs; hipcc -> zpack_amd/dev/${ENC_VARIANT_SO:-ab_enctag.so}.
# Result: profiles/r05/r05_enc_tag_experiment.txt
out=gpurun_out/r05enc; mkdir -p $out
for v in head tag head tag; do
  if [ $v = tag ]; then export ZPACK_AMD_CODEC_SO=$PWD/zpack_amd/dev/${ENC_VARIANT_SO:-ab_enctag.so}; else unset ZPACK_AMD_CODEC_SO; fi
  timeout -k 10 400 python bench.py --workload c5_zstd1_1m --steps 3 --warmup 1 --no-cpu > $out/$v.json 2> $out/$v.err || { echo "$v failed"; tail -3 $out/$v.err; exit 1; }
  python3 - <<PY
import json
d=json.loads(open("$out/$v.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$v: %.1f GiB/s  %.2f ms/step  kernel %s  ratio %s  parity %s" % (d["value"], d["ms_per_step"], [round(x,1) for x in r.get("stage_ms",[])], d["config"].get("ratio"), all(v for v in d["parity"].values() if v is not None)))
PY
done
