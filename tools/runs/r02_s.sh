#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_s
timeout 900 python tools/ab_key.py 7 2>&1 | tee gpurun_out/r02_s/ab_streamk.txt
