#!/bin/bash
# coarse32_kernel v2 (32x32x16 MFMA, explicit LDS prefetch a pair ahead): parity, fuzz on small dims, A/B key 11
cd /root/repo
O=gpurun_out/r02_aw; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-300
VQ_FUZZ_DIMS=8,16,24,32 timeout 300 python tools/fuzz_vs_exact.py 120 51 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz.log | cut -c1-300
timeout 600 python tools/ab_key.py 11 2>&1 | grep -v amdgpu.ids | grep "D=  32" | tee $O/ab_wide.txt
