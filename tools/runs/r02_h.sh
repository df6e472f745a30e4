#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_h
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r02_h/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r02_h/pytest.log | cut -c1-250
timeout 600 python tools/bench_shapes.py 2>&1 | tee gpurun_out/r02_h/shapes.txt
timeout 600 python tools/fuzz_vs_exact.py 120 2>&1 | tail -5 | tee gpurun_out/r02_h/fuzz.txt
