#!/bin/bash
# round 5: the XXH3 chain on the scalar unit (eight waves, one accumulator each) against the vector chain (one wave): big-entry tests, then one 256 MiB frame
out=gpurun_out/r05cs; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_big_entries.py -m gpu -x -q > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc -eq 0 ] || exit $rc
for v in scalar vector scalar vector; do
  if [ $v = vector ]; then export ZPACK_AMD_CODEC_SO=$PWD/zpack_amd/dev/ab_chainv.so; else unset ZPACK_AMD_CODEC_SO; fi
  echo "== chain: $v"; timeout -k 10 300 python3 tools/big_frame_rate.py 256 2 lz4 2>&1 | grep -v amdgpu.ids | grep "^text \|^byte" | cut -c1-175
done | tee $out/ab.txt
unset ZPACK_AMD_CODEC_SO
root=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/$out/trace -o t --output-format csv -- python3 $root/tools/big_frame_rate.py 256 2 lz4 > $root/$out/rate.txt 2> $root/$out/err.txt
python3 - <<PY
import csv,glob
for f in glob.glob("$root/$out/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Name"].split("(")[0]
        if "xxh3" in n: print("   %-24s calls %5s total %.2f ms avg %.3f ms" % (n[-24:], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6))
PY
