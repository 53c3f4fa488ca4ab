#!/bin/bash
# encoder A/B on one box: the encoder tests, then C5 on text (6000 entries, the configuration of profiles/r03/r03_e_encoder_phase_ablation.txt) and on the mix
mkdir -p gpurun_out/enc
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "write or c5 or encoder or level or pack or roundtrip" > gpurun_out/enc/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/enc/pytest.log
for so in zpack_amd/libzpk_codec.so $(ls zpack_amd/abl_*.so 2>/dev/null); do
  for args in "--mix 0 --entries 6000" ""; do
    ZPACK_AMD_CODEC_SO=$PWD/$so timeout -k 10 400 python3 bench.py --workload c5_zstd1_1m --steps 3 --warmup 1 --no-cpu $args > gpurun_out/enc/line.json 2> gpurun_out/enc/err.txt || { echo "bench failed: $so $args"; tail -5 gpurun_out/enc/err.txt; }
    python3 - "$so" "$args" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/enc/line.json").read().strip().splitlines()[-1])
print("%-34s %-24s GiB/s %.1f ms/step %.2f stage_ms %s ratio %s parity %s" % (sys.argv[1].split("/")[-1], sys.argv[2], d["value"], d["ms_per_step"], [round(x, 2) for x in d["roofline"]["stage_ms"]], d["config"].get("comp_ratio"), d.get("parity")))
PY
  done
done | tee gpurun_out/enc/ab.txt
