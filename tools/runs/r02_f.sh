#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_f
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r02_f/pytest.log 2>&1; echo "pytest rc=$?"
tail -4 gpurun_out/r02_f/pytest.log | cut -c1-250
timeout 600 python tools/graph_vs_eager.py 2>&1 | tee gpurun_out/r02_f/graph.txt
