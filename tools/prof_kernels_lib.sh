#!/bin/bash
# Per-kernel averages (rocprofv3 --kernel-trace --stats) of bench.py's default workload for one build of the library.
# usage (on the GPU box): tools/prof_kernels_lib.sh <tag> <path of libvqhip.so | shipped> [pattern of kernel names to print]
tag=$1; lib=$2; pat=${3:-.}
[ "$lib" != shipped ] && export VQHIP_LIB=/root/repo/$lib
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$tag -- python3 /root/repo/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-verify --min-seconds 0 > /root/repo/gpurun_out/$tag.json 2> /root/repo/gpurun_out/$tag.err
python3 - /root/repo/gpurun_out/$tag "$pat" <<'PY'
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if re.search(sys.argv[2], r['Name']):
        print(f"  {r['Name'][:70]:72s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f}")
PY
