#!/bin/bash
# usage (on the GPU box): tools/prof_train.sh <tag> [N]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$tag -- python3 /root/repo/tools/bench_train.py "$@" > /root/repo/gpurun_out/$tag.log 2>&1
tail -1 /root/repo/gpurun_out/$tag.log
python3 - /root/repo/gpurun_out/$tag <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print(f"  {r['Name'][:70]:72s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f}  {r['Percentage']}%")
PY
