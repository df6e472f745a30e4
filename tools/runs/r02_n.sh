#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_n
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r02_n/pytest.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/r02_n/pytest.log | cut -c1-300
timeout 600 python tools/ab_key.py 6 2>&1 | tee gpurun_out/r02_n/ab_decide.txt
timeout 600 python tools/bench_shapes.py 2>&1 | tee gpurun_out/r02_n/shapes.txt
timeout 400 python tools/fuzz_vs_exact.py 150 11 > gpurun_out/r02_n/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -4 gpurun_out/r02_n/fuzz.txt
