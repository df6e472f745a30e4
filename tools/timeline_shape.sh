#!/bin/bash
# Per-launch timeline of one encode (run on the GPU box): tools/timeline_shape.sh <tag> <opening kernel substring> N K D L2|Cosine
# -> gpurun_out/<tag>_timeline.txt (kernel, duration, gap) of the second-to-last call.  Env VQ_PROF_ENCODE / VQ_PROF_BF16: prof_shape.py
tag=$1; open=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/$tag -- python3 /root/repo/tools/prof_shape.py "$@" 30 > /root/repo/gpurun_out/$tag.log 2>&1
python3 /root/repo/tools/step_timeline.py /root/repo/gpurun_out/$tag "$open" > /root/repo/gpurun_out/${tag}_timeline.txt 2>&1
cat /root/repo/gpurun_out/${tag}_timeline.txt
