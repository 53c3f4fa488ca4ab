#!/bin/bash
# tools/r5_trace.sh <label> <bench args...>: rocprofv3 kernel trace + stats of one bench run, summary printed
label=$1; shift
root=$PWD; out=$root/gpurun_out/r05t_$label; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out/trace -o t --output-format csv -- python3 $root/bench.py "$@" --no-cpu > $out/bench.json 2> $out/bench.err; echo "rc=$?"
f=$(find $out/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats.csv && column -s, -t < $f | cut -c1-150 | head -14
