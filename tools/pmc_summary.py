"""Summarise tools/pmc.sh output: per kernel, the mean of every counter over its dispatches.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (MI355X_MICROARCH.md, HBM section); they are
converted to bytes here.  Usage: python tools/pmc_summary.py <dir> [--json out.json]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def summarise(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("zpk::", "").split("<")[0].strip()      # k_encode<12> and <14> are one kernel here
                if not name.startswith("k_") and "zpk" not in name:
                    continue
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, cs in acc.items():
        out[k] = {}
        for c, v in cs.items():
            # all FULL launches of a kernel do the same work; bench.py also launches the kernels on its small copy-ceiling batch (and
            # on batches that hold no entry of the kernel's method): those dispatches, far below the largest, are left out of the mean
            top = max(v)
            v = [x for x in v if x >= 0.5 * top] if top > 0 else v
            m = sum(v) / len(v)
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                m *= 1024.0
                c += "_bytes"
            out[k][c] = m
        out[k]["dispatches"] = max(len(v) for v in cs.values())
    return out


if __name__ == "__main__":
    s = summarise(sys.argv[1])
    if "--json" in sys.argv:
        doc = {"kernels": s, "source": "rocprofv3 --pmc, one pass per counter group (tools/pmc.sh)",
               "units": "FETCH_SIZE/WRITE_SIZE converted KiB -> bytes, uncorrected; other counters raw, mean per dispatch"}
        if "--entries" in sys.argv:
            doc["entries_per_gpu"] = int(sys.argv[sys.argv.index("--entries") + 1])
        if "--workload" in sys.argv:
            doc["workload"] = sys.argv[sys.argv.index("--workload") + 1]
        # identity of the kernel sources the counters were measured on: bench.py quotes the traffic only for this very code
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import csrc_sha1
        doc["csrc_sha1"] = csrc_sha1()
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(doc, fh, indent=1, sort_keys=True)
    for k in sorted(s):
        print(k)
        for c in sorted(s[k]):
            print("    %-32s %.6g" % (c, s[k][c]))
