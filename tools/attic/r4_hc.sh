#!/bin/bash
# level >= 3 encoder: 2^14 slots (32 KiB, 4 workgroups per CU) against 2^13 (16 KiB, 8 per CU) — rate and ratio per class (enc_bench, level 3)
for so in zpack_amd/abl_hc*.so; do
  echo "== $(basename $so .so)"
  ZPACK_AMD_CODEC_SO=$PWD/$so timeout -k 10 400 python3 tools/enc_bench.py 2000 1048576 3 2>&1 | grep -E "^(text|records|random|runs) "
done
