import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'pre_kernel<0, true' in r['Kernel_Name']]
i0, i1 = marks[-3], marks[-2]
prev = None
for r in rows[i0:i1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(e - s) / 1e3:7.2f} us gap {((s - prev) / 1e3 if prev else 0.0):6.2f} grid {r['Grid_Size_X']:>8s} wg {r['Workgroup_Size_X']:>5s}  {r['Kernel_Name'][:230]}")
    prev = e
print('wall', (int(rows[i1]['Start_Timestamp']) - int(rows[i0]['Start_Timestamp'])) / 1e3)
