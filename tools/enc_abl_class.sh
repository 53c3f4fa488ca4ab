#!/bin/bash
# encoder phase ablation on one corpus class (developer): tools/enc_abl_class.sh <mix 0..3> [entries]   (needs zpack_amd/abl_*.so from tools/abl.sh)
mkdir -p gpurun_out/enc
for so in zpack_amd/libzpk_codec.so $(ls zpack_amd/abl_*.so 2>/dev/null); do
  ZPACK_AMD_CODEC_SO=$PWD/$so timeout -k 10 400 python3 bench.py --workload c5_zstd1_1m --steps 2 --warmup 1 --no-cpu --mix $1 --entries ${2:-6000} > gpurun_out/enc/line.json 2> gpurun_out/enc/err.txt || { echo "bench failed: $so"; tail -3 gpurun_out/enc/err.txt; }
  python3 - "$so" "$1" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/enc/line.json").read().strip().splitlines()[-1])
print("%-28s mix %s GiB/s %.1f stage_ms %s ratio %.4f" % (sys.argv[1].split("/")[-1], sys.argv[2], d["value"], [round(x, 2) for x in d["roofline"]["stage_ms"]], d["config"].get("comp_ratio")))
PY
done | tee gpurun_out/enc/abl_class_$1.txt
