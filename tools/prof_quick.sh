#!/bin/bash
# kernel-trace stats of a bench run + one PMC pass of instruction counters; summaries under gpurun_out/prof_quick/
out=$PWD/gpurun_out/prof_quick; rm -rf $out; mkdir -p $out
args=${@:---entries 30000 --steps 3 --warmup 1 --no-cpu}
root=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $root/bench.py $args > $out/kt.log 2>&1
echo "kernel-trace rc=$?"
f=$(find $out/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -d, -f1-8 "$f" | head -12
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out/pmc1 -o p --output-format csv -- python3 $root/bench.py $args > $out/pmc1.log 2>&1
echo "pmc rc=$?"
python3 - <<PY
import csv,glob,collections
for f in glob.glob("$out/pmc1/**/*counter_collection.csv", recursive=True):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        n=len(cnt[k]); print(k[:40], "dispatches",n, {c: "%.3g"%(x/n) for c,x in sorted(v.items())})
PY
