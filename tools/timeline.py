#!/usr/bin/env python3
"""Print one iteration of a rocprofv3 kernel trace as a timeline (kernel, duration, gap to the previous kernel).
usage: timeline.py <dir with *_kernel_trace.csv> <kernel-name-prefix that starts an iteration> [which iteration from the end]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
marks = [i for i, n in enumerate(names) if n.startswith(sys.argv[2]) or ('void ' + sys.argv[2]) in n[:len(sys.argv[2]) + 6]]
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
i0, i1 = marks[-back - 1], marks[-back]
prev = None
busy = 0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    busy += e - s
    print(f"{r['Kernel_Name'].split('(')[0][-60:]:62s} {(e - s) / 1e3:8.2f} us   gap {gap:7.2f}")
    prev = e
wall = (int(rows[i1]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3
print(f'iteration: {i1 - i0} kernels, wall {wall:.1f} us, kernel time {busy / 1e3:.1f} us')
