#!/usr/bin/env python3
"""Developer: timeline of one large LZ4 frame from a rocprofv3 kernel + memory-copy trace (tools/r5_pjtrace.sh).  usage: pj_timeline.py DIR [frame_index]"""
import csv, sys, glob
d = sys.argv[1]; which = int(sys.argv[2]) if len(sys.argv) > 2 else 4
k = [r for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
m = [r for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True) for r in csv.DictReader(open(f))]
ev = []
for r in k: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].split('::')[-1][:18]))
for r in m: ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r['Direction'].replace('MEMORY_COPY_', '')))
ev.sort()
idx = [i for i, e in enumerate(ev) if 'k_pj_scan' in e[2]]
i0 = idx[which]; j = i0
while j > 0 and (ev[j - 1][2].startswith('COPY') or 'copyBuffer' in ev[j - 1][2]) and ev[i0][0] - ev[j - 1][0] < 20e6: j -= 1
t0 = ev[j][0]; jumps = 0; jt = 0
end = idx[which + 1] if which + 1 < len(idx) else len(ev)
for e in ev[j:end]:
    if 'k_pj_jump' in e[2]: jumps += 1; jt += e[1] - e[0]; continue
    if 'fillBuffer' in e[2] or 'illBuffer' in e[2]: continue
    print("%9.3f ms +%8.3f  %s" % ((e[0] - t0) / 1e6, (e[1] - e[0]) / 1e6, e[2]))
print("%d jump launches, %.3f ms" % (jumps, jt / 1e6))
