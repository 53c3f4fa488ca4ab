#!/bin/bash
# tools/r5_pmc.sh <label> <kernel-substring> <bench args...>: instruction / wait counters of one kernel
label=$1; pat=$2; shift 2
root=$PWD; out=$root/gpurun_out/r05p_$label; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp -d $out/g$i -o p --output-format csv -- python3 $root/bench.py "$@" --no-cpu > $out/g$i.log 2>&1; echo "group $i rc=$?"
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
        print(k[:20], {c: "%.4g"%max(x) for c,x in sorted(v.items())})
PY
grep -o '"lz4_chunks_from_records": [0-9]*, "lz4_chunk_records_rejected": [0-9]*' $out/g1.log | tail -1
