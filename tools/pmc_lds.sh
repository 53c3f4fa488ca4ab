#!/bin/bash
out=$PWD/gpurun_out/pmc_lds; rm -rf $out; mkdir -p $out
args=${@:---entries 30000 --steps 3 --warmup 1 --no-cpu --mix 0}
root=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS[A-Z_]*\|SQ_WAIT_INST_LDS\|SQ_INSTS_LDS[A-Z_]*\|SQ_ACTIVE_INST_LDS" | sort -u | tr '\n' ' ' > $out/lds_counters.txt; cat $out/lds_counters.txt; echo
i=0
for grp in "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp -d $out/g$i -o p --output-format csv -- python3 $root/bench.py $args > $out/g$i.log 2>&1
  echo "group $i rc=$?"
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$out/g*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "lz4" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        n=len(cnt[k]); print(k[:20], {c: "%.3g"%(x/n) for c,x in sorted(v.items())})
PY
