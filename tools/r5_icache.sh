#!/bin/bash
# round 5: is k_lz4_wave's sensitivity to code size an instruction-cache effect?  SQC instruction-cache counters of two builds (v0 = round-4 code, v2 = + 400 lines of guarded code)
out=$PWD/gpurun_out/r05i; rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o -E "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_LEVEL|DCACHE)[A-Z0-9_]*)\b" | sort -u > $out/counters.txt
cat $out/counters.txt | tr '\n' ' '; echo
for v in 0 2; do
  export ZPACK_AMD_CODEC_SO=$root/zpack_amd/dev/ab_v$v.so
  i=0
  for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp -d $out/v${v}_g$i -o p --output-format csv -- python3 $root/bench.py --mix 0 --entries 30000 --steps 3 --warmup 1 --no-cpu > $out/v${v}_g$i.log 2>&1
    echo "v$v group $i rc=$?"
  done
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$out/v*_g*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("zpk::","")
        if "k_lz4_wave" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        n=len(cnt[k]); print(f.split("/r05i/")[1].split("/")[0], k[:20], {c: "%.4g"%(x/n) for c,x in sorted(v.items())})
PY
