#!/bin/bash
# round 4: the two stage-2 executors side by side: k of every 8 work-list slots through the LDS window, the rest over the output slot
tag=${1:-x}; out=$PWD/gpurun_out/r4_mix_$tag; rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
args="--entries 100000 --steps 5 --warmup 2 --no-cpu"
export ZPK_BENCH_LZ4_TWO=never
timeout -k 10 200 python3 $root/bench.py $args > $out/bench_one.json 2> $out/bench_one.err
python3 -c "
import json
d=json.loads(open('$out/bench_one.json').read().strip().splitlines()[-1]); r=d['roofline']; print('one-kernel: %.1f GiB/s kernel_ms %.3f parity %s' % (d['value'], r['kernel_ms'], d['parity']['all_ranks']))" || tail -3 $out/bench_one.err
unset ZPK_BENCH_LZ4_TWO
for k in 2 3 4 5 6; do
  export ZPK_BENCH_LZ4_EXEC_WINDOW=$((k+2))
  timeout -k 10 200 python3 $root/bench.py $args > $out/bench_$k.json 2> $out/bench_$k.err
  python3 -c "
import json
d=json.loads(open('$out/bench_$k.json').read().strip().splitlines()[-1]); r=d['roofline']; p=r.get('lz4_parse_ms') or 0.0
print('window %d of 8: %.1f GiB/s kernel_ms %.3f parse_ms %.3f rest_ms %.3f parity %s' % ($k, d['value'], r['kernel_ms'], p, r['kernel_ms']-p, d['parity']['all_ranks']))" || tail -3 $out/bench_$k.err
done
