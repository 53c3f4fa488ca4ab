#!/bin/bash
# round 4: non-temporal loads for far match sources (SEQ_FAR_NT = offset threshold in bytes): one-kernel path and two-stage slot path
out=$PWD/gpurun_out/r4_farnt; rm -rf $out; mkdir -p $out
root=$PWD
args="--entries 100000 --steps 6 --warmup 2 --no-cpu"
for v in base nt2048 nt4096 nt8192; do
  if [ $v = base ]; then unset ZPACK_AMD_CODEC_SO; else export ZPACK_AMD_CODEC_SO=$root/zpack_amd/dev/libzpk_codec_$v.so; fi
  for mode in one g; do
    unset ZPK_BENCH_LZ4_TWO ZPK_BENCH_LZ4_EXEC_WINDOW
    [ $mode = g ] && export ZPK_BENCH_LZ4_TWO=always ZPK_BENCH_LZ4_EXEC_WINDOW=0
    timeout -k 10 200 python3 bench.py $args > $out/${v}_$mode.json 2> $out/${v}_$mode.err
    python3 -c "
import json
d=json.loads(open('$out/${v}_$mode.json').read().strip().splitlines()[-1]); r=d['roofline']; p=r.get('lz4_parse_ms') or 0.0
print('$v $mode: %.1f GiB/s kernel_ms %.3f (parse %.3f, rest %.3f) parity %s' % (d['value'], r['kernel_ms'], p, r['kernel_ms']-p, d['parity']['all_ranks']))" || tail -2 $out/${v}_$mode.err
  done
done
cd /tmp && export TMPDIR=/tmp
for v in base nt4096; do
  if [ $v = base ]; then unset ZPACK_AMD_CODEC_SO; else export ZPACK_AMD_CODEC_SO=$root/zpack_amd/dev/libzpk_codec_$v.so; fi
  unset ZPK_BENCH_LZ4_TWO ZPK_BENCH_LZ4_EXEC_WINDOW
  timeout -k 10 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum -d $out/pmc_$v -o p --output-format csv -- python3 $root/bench.py --entries 100000 --steps 3 --warmup 1 --no-cpu > $out/pmc_$v.log 2>&1
  echo "== $v (one-kernel path)"; python3 $root/tools/pmc_summary.py $out/pmc_$v | grep -A6 "^k_lz4_wave"
done
