#!/bin/bash
cd /root/repo
O=gpurun_out/r02_ar; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/b32 -- python3 /root/repo/bench.py --images 32 --no-cpu-baseline --steps 30 --warmup 5 > /root/repo/$O/b32.json 2> /root/repo/$O/b32.err
cd /root/repo
python3 tools/timeline.py $O/b32 pre_kernel 20 > $O/timeline.txt 2>&1; cut -c1-110 $O/timeline.txt
