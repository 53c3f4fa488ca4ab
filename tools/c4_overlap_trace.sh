#!/bin/bash
# evidence for DESIGN 4.2b: kernel start / end times of one mixed batch (rocprofv3 kernel trace) — k_lz4_wave runs INSIDE the Zstandard stages
out=$PWD/gpurun_out/r03/c4_trace; rm -rf $out; mkdir -p $out
root=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace -d $out -o t --output-format csv -- python3 $root/bench.py --workload c4_mixed --steps 1 --warmup 1 --no-cpu > $out/bench.json 2> $out/err.txt; echo "trace rc=$?")
python3 - <<PY
import csv, glob
f = glob.glob("$out/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_lz4_wave", "k_zstd_fse", "k_zstd_exec", "k_zstd(", "k_classify", "k_lz4_retry"))]
S = lambda r: int(r["Start_Timestamp"]); E = lambda r: int(r["End_Timestamp"])
last = max((r for r in keep if "k_zstd_exec" in r["Kernel_Name"] and E(r) - S(r) > 5_000_000), key=E)      # the last full batch's execute stage
t0 = max(S(r) for r in keep if "k_classify" in r["Kernel_Name"] and S(r) < S(last))                      # ... and its first kernel
sel = [r for r in keep if t0 <= S(r) <= E(last)]
out = ["# one mixed batch of 125 000 entries (bench.py --workload c4_mixed --steps 1 --warmup 1 under rocprofv3 --kernel-trace; tools/c4_overlap_trace.sh):",
       "# start / end in ms from the batch's first kernel, and the hardware queue.  k_lz4_wave (side stream, low priority) runs INSIDE k_zstd_fse."]
for r in sorted(sel, key=S):
    out.append("%-14s start %8.2f  end %8.2f  (%6.2f ms)  queue %s" % (r["Kernel_Name"].split("(")[0].replace("zpk::", "")[:14], (S(r) - t0) / 1e6, (E(r) - t0) / 1e6, (E(r) - S(r)) / 1e6, r.get("Queue_Id", "?")))
open("$PWD/gpurun_out/r03/r03_c4_mixed_kernel_timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
