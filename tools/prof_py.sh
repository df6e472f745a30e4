#!/bin/bash
# usage (on the GPU box): tools/prof_py.sh <tag> <script.py> [args...]  -> per-kernel stats of any script under tools/
tag=$1; shift; script=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$tag -- python3 /root/repo/$script "$@" > /root/repo/gpurun_out/$tag.log 2>&1
python3 - /root/repo/gpurun_out/$tag <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print(f"  {r['Name'][:64]:66s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f}  {r['Percentage']}%")
PY
