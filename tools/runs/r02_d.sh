#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_d
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r02_d/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r02_d/pytest.log | cut -c1-250
