#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_m
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r02_m/cvq -- python3 /root/repo/bench.py --workload cvq --no-cpu-baseline --steps 20 --warmup 5 > /root/repo/gpurun_out/r02_m/cvq.json 2> /root/repo/gpurun_out/r02_m/cvq.err
cd /root/repo
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r02_m/cvq/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
bw = [i for i, n in enumerate(names) if 'vq_backward_kernel' in n]
# the graphed steps are the last ones: take an iteration near the end
i0, i1 = bw[-4] + 1, bw[-3] + 1
prev = None; busy = 0
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0
    busy += e - s
    print(f"{r['Kernel_Name'].split('(')[0][-56:]:58s} {(e-s)/1e3:7.2f} gap {gap:7.2f}")
    prev = e
print('kernels', i1 - i0, 'wall', (int(rows[i1]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3, 'busy', busy / 1e3)
PY
