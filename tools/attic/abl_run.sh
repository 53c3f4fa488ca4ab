#!/bin/bash
# time every zpack_amd/abl_*.so on one bench workload: tools/abl_run.sh [bench args]
args=${@:---entries 30000 --steps 3 --warmup 1 --no-cpu --mix 0}
mkdir -p gpurun_out
for so in zpack_amd/abl_*.so; do
  n=$(basename $so .so)
  ZPACK_AMD_CODEC_SO=$PWD/$so timeout -k 10 300 python bench.py $args 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-14s'%'$n', 'GiB/s %.1f'%d['value'], 'ms/step %.3f'%d['ms_per_step'], 'stage_ms', [round(x,3) for x in d['roofline']['stage_ms']], 'ratio', d['config'].get('comp_ratio'), 'parity', d['parity'].get('all_ranks'))
"
done | tee gpurun_out/abl_run.txt
