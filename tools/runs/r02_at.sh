#!/bin/bash
cd /root/repo
O=gpurun_out/r02_at; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/c3 -- python3 /root/repo/tools/c3_once.py > /root/repo/$O/c3.log 2>&1
cd /root/repo
python3 tools/timeline.py $O/c3 pre_kernel 5 2>&1 | cut -c1-110
