#!/usr/bin/env python3
"""One whole step of a rocprofv3 kernel trace as a timeline (kernel, duration, gap to the previous kernel).
usage: step_timeline.py <dir with *_kernel_trace.csv> <substring of the kernel that opens a step> [which step from the end]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
i0, i1 = marks[-back - 1], marks[-back]
prev, busy = None, 0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    busy += e - s
    print(f"{r['Kernel_Name'].split('(')[0][-66:]:68s} {(e - s) / 1e3:8.2f} us   gap {((s - prev) / 1e3 if prev else 0.0):7.2f}")
    prev = e
print(f"step: {i1 - i0} kernels, wall {(int(rows[i1]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us")
