#!/bin/bash
# tools/pmc_traffic.sh <kernel-substring> <bench args...>: FETCH_SIZE / WRITE_SIZE (KiB, raw) of one kernel, for the codec .so in
# $ZPACK_AMD_CODEC_SO (default: the in-tree build); two rocprofv3 --pmc passes, never combined with trace domains
pat=$1; shift
out=$PWD/gpurun_out/pmc_traffic; rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp -d $out/g$i -o p --output-format csv -- python3 $root/bench.py "$@" > $out/g$i.log 2>&1
  echo "group $i rc=$?"
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$out/g*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("zpk::","")
        if "$pat" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        o={}
        for c,x in sorted(v.items()):
            top=max(x); x=[y for y in x if y>=0.5*top] if top>0 else x
            m=sum(x)/len(x)
            o[c]="%.4g"%(m*1024 if c in ("FETCH_SIZE","WRITE_SIZE") else m)
        print("${ZPACK_AMD_CODEC_SO##*/}", k[:20], o)
PY
