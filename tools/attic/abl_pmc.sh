#!/bin/bash
# instruction counters of every zpack_amd/abl_*.so on one bench workload (developer): tools/abl_pmc.sh <kernel-substring> [bench args]
pat=$1; shift
args=${@:---entries 30000 --steps 2 --warmup 1 --no-cpu --mix 0}
out=$PWD/gpurun_out/abl_pmc; rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
for so in $root/zpack_amd/abl_*.so; do
  n=$(basename $so .so)
  export ZPACK_AMD_CODEC_SO=$so
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out/$n -o p --output-format csv -- python3 $root/bench.py $args > $out/$n.log 2>&1
  echo "$n rc=$?"
done
python3 - <<PY | tee $out/summary.txt
import csv,glob,collections,os
for d in sorted(glob.glob("$out/abl_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0].replace("zpk::","")
            if "$pat" not in k: continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            print("%-12s %-14s"%(os.path.basename(d),k[:14]), {c: "%.4g"%max(x) for c,x in sorted(v.items())})
PY
