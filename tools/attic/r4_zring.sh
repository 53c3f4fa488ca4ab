#!/bin/bash
# round 4: the Zstandard execute stage with a larger LDS output ring (how many match sources stay on chip vs. workgroups per CU)
out=$PWD/gpurun_out/r4_zring; rm -rf $out; mkdir -p $out
for v in base r6k r8k r12k; do
  if [ $v = base ]; then unset ZPACK_AMD_CODEC_SO; else export ZPACK_AMD_CODEC_SO=$PWD/zpack_amd/dev/libzpk_codec_$v.so; fi
  timeout -k 10 250 python3 bench.py --workload c3_zstd_256k --entries 40000 --steps 4 --warmup 2 --no-cpu > $out/c3_$v.json 2> $out/c3_$v.err
  python3 -c "
import json
d=json.loads(open('$out/c3_$v.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$v: %.1f GiB/s stage_ms %s parity %s' % (d['value'], [round(x,2) for x in r['stage_ms']], d['parity']['all_ranks']))" || tail -3 $out/c3_$v.err
done
