#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_g
tools/prof_py.sh r02_g/cvq tools/bench_cvq.py 3072 > gpurun_out/r02_g/cvq_stats.txt 2>&1
python3 tools/timeline.py gpurun_out/r02_g/cvq cb_stats_kernel 3 2>&1 | tee gpurun_out/r02_g/cvq_timeline.txt
grep "ms per" gpurun_out/r02_g/cvq.log
