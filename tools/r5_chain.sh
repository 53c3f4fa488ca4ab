#!/bin/bash
# round 5: the XXH3 chain kernel, three formulations of one step (XS_CHAIN_VARIANT), same box: kernel time per 256 MiB span
root=$PWD; out=$root/gpurun_out/r05chain; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in 0 1 2; do
  export ZPACK_AMD_CODEC_SO=$root/zpack_amd/dev/ab_chain$v.so
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/v$v -o t --output-format csv -- python3 $root/tools/big_frame_rate.py 256 4 > $out/rate$v.txt 2> $out/err$v.txt || { echo "variant $v failed"; tail -5 $out/err$v.txt; exit 1; }
  grep "^text " $out/rate$v.txt
  python3 - <<PY
import csv,glob
for f in glob.glob("$out/v$v/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"].split("(")[0]
        if "xxh3_chain" in n: print("   variant $v  %-18s calls %5s total %.2f ms -> %.2f ms per 256 MiB span" % (n, r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["TotalDurationNs"])/1e6/12))
PY
done
