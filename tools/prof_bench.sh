#!/bin/bash
# usage (on the GPU box): tools/prof_bench.sh <tag> [bench.py args]
# ONE command gives both artefacts the judge compares: bench.py's JSON line (HIP-event time of the proposal kernel)
# and rocprofv3's per-kernel summary of the very same process.  Many steps, so the clock-ramp launches do not skew
# the profiler's average.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$tag -- python3 /root/repo/bench.py --steps 200 --warmup 10 "$@" > /root/repo/gpurun_out/$tag.json 2> /root/repo/gpurun_out/$tag.err
python3 - /root/repo/gpurun_out/$tag <<'PY'
import csv,sys,glob,json
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:12]:
    print(f"  {r['Name'][:60]:62s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f}  {r['Percentage']}%")
j=json.loads(open(sys.argv[1]+'.json').read().strip().splitlines()[-1])
k=[r for r in rows if 'coarse_kernel' in r['Name']][0]
print('bench.py  : value %.1f M tokens/s, kernel_ms (HIP events) %.4f' % (j['value']/1e6, j['roofline']['kernel_ms']))
print('rocprofv3 : coarse_kernel average %.4f ms over %s launches' % (float(k['AverageNs'])/1e6, k['Calls']))
PY
