#!/usr/bin/env python3
"""One whole training step of a rocprofv3 kernel trace as a timeline: the launches between two consecutive occurrences of the
kernel that ENDS a step (the backward kernel), i.e. forward first.
usage: step_timeline_after.py <dir with *_kernel_trace.csv> <substring of the step's last kernel> [which step from the end]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
i0, i1 = marks[-back - 1] + 1, marks[-back] + 1
prev, busy = int(rows[i0 - 1]['End_Timestamp']), 0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    print(f"{r['Kernel_Name'].split('(')[0][-72:]:74s} {(e - s) / 1e3:8.2f} us   gap {(s - prev) / 1e3:7.2f}")
    prev = e
wall = (int(rows[i1 - 1]['End_Timestamp']) - int(rows[i0 - 1]['End_Timestamp'])) / 1e3
print(f"step: {i1 - i0} kernels, wall {wall:.1f} us, kernel time {busy / 1e3:.1f} us, gaps {wall - busy / 1e3:.1f} us")
