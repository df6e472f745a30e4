#!/bin/bash
# stage barrier that keeps the newest stage's LDS-DMA in flight: parity, fuzz, A/B against the vmcnt(0) build
cd /root/repo
O=gpurun_out/r02_ae; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
timeout 400 python tools/fuzz_vs_exact.py 150 37 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log | cut -c1-300
E=build/exp
for shape in "100352 8192 32 Cosine" "65536 8192 32 Cosine" "524288 16384 8 L2" "65536 8192 64 L2" "524288 16384 256 L2" "8192 16384 256 L2" "3072 16384 256 Cosine" "65536 8192 128 Cosine"; do
timeout 600 python tools/exp_shape.py $shape shipped $E/libvqhip_fullwait.so 2>&1 | grep -v amdgpu.ids | tee -a $O/shapes.txt
done
