#!/bin/bash
# tools/pmc_any.sh <kernel-substring> <bench args...>: LDS/issue PMC groups for one kernel
pat=$1; shift
out=$PWD/gpurun_out/pmc_any; rm -rf $out; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $grp -d $out/g$i -o p --output-format csv -- python3 $root/bench.py "$@" > $out/g$i.log 2>&1
  echo "group $i rc=$?"
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$out/g*/**/*counter_collection.csv", recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("zpk::","")
        if "$pat" not in k: continue
        agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k,v in agg.items():
        n=len(cnt[k]); print(k[:20], {c: "%.3g"%(x/n) for c,x in sorted(v.items())})
PY
