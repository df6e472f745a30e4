#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_r
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r02_r/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r02_r/pytest.log | cut -c1-300
timeout 900 python tools/ab_key.py 7 2>&1 | tee gpurun_out/r02_r/ab_streamk.txt
VQ_FUZZ_DIMS=8,16,32,24 timeout 400 python tools/fuzz_vs_exact.py 120 21 > gpurun_out/r02_r/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -4 gpurun_out/r02_r/fuzz.txt
