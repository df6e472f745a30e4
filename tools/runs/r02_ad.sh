#!/bin/bash
# balanced tiles per workgroup (key 10): parity, fuzz, A/B
cd /root/repo
O=gpurun_out/r02_ad; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
timeout 400 python tools/fuzz_vs_exact.py 200 31 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log | cut -c1-300
timeout 600 python tools/ab_key.py 10 2>&1 | grep -v amdgpu.ids | tee $O/ab_balance.txt
