#!/bin/bash
# Per-launch timeline of one GRAPH-REPLAYED training step at a per-rank shape (run on the GPU box):
#   tools/timeline_train_graphed.sh <tag> <shape name> <substring of the step's last kernel>  -> gpurun_out/<tag>_timeline.txt
tag=$1; name=$2; last=$3
cd /tmp && export TMPDIR=/tmp
VQ_TRAIN_STEPS=20 VQ_TRAIN_SETTLE=150 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/$tag -- python3 /root/repo/tools/bench_train_shapes.py $name > /root/repo/gpurun_out/$tag.log 2>&1
python3 /root/repo/tools/step_timeline_after.py /root/repo/gpurun_out/$tag "$last" > /root/repo/gpurun_out/${tag}_timeline.txt 2>&1
cat /root/repo/gpurun_out/${tag}_timeline.txt
