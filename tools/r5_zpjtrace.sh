#!/bin/bash
# kernel trace of one large Zstandard frame through the block-parallel path
root=$PWD; out=$root/gpurun_out/r05zpj; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $out/trace -o t --output-format csv -- python3 $root/tools/big_frame_rate.py 256 2 zstd 3 > $out/rate.txt 2> $out/err.txt; echo "rc=$?"
grep "text " $out/rate.txt | head -2
python3 $root/tools/pj_timeline.py $out/trace 4 > $out/timeline.txt
python3 - <<PY
import csv,glob
for f in glob.glob("$out/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"].split("(")[0]
        if "pj" in n or "xxh3" in n or "zstd" in n: print("   %-24s calls %5s total %.2f ms avg %.3f ms" % (n[-24:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
