#!/bin/bash
# usage (on the GPU box): tools/prof_breakdown.sh <tag> [env assignments...]   -> gpurun_out/<tag>/ kernel stats
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
env "$@" true
for kv in "$@"; do export "$kv"; done
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$tag -- python3 /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /root/repo/gpurun_out/$tag.log 2>&1
python3 - /root/repo/gpurun_out/$tag <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows[:13]:
    if 'at::native' in r['Name'] and 'FillFunctor<double>' not in r['Name']: continue
    print(f"  {r['Name'][:44]:46s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
grep -o '"value": [0-9.]*' /root/repo/gpurun_out/$tag.log
