#!/bin/bash
# filtered small-D epilogue: parity suite, A/B against the unfiltered kernel in one process, C3 profile + PMC
cd /root/repo
mkdir -p gpurun_out/r02_c
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_c/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r02_c/pytest.log
timeout 600 python tools/ab_filter.py > gpurun_out/r02_c/ab.txt 2>&1; echo "ab rc=$?"; cat gpurun_out/r02_c/ab.txt
timeout 600 tools/prof_shape.sh r02_c/c3_stats 100352 8192 32 Cosine
timeout 900 tools/pmc_shape.sh r02_c/c3_pmc 100352 8192 32 Cosine
