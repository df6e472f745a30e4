#!/bin/bash
# NOAUX proposal kernel (cosine/dot, D <= 32): parity, fuzz incl. padded last stage, A/B key 8, workgroup-shape variants
cd /root/repo
O=gpurun_out/r02_y; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-200
VQ_FUZZ_DIMS=8,16,32,32 timeout 300 python tools/fuzz_vs_exact.py 150 7 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
timeout 600 python tools/ab_key.py 8 2>&1 | grep -v amdgpu.ids | tee $O/ab_noaux.txt
E=build/exp
for shape in "100352 8192 32 Cosine" "65536 8192 32 Cosine" "131072 8192 32 Cosine" "524288 16384 8 L2"; do
timeout 600 python tools/exp_shape.py $shape shipped $E/libvqhip_t4w4.so $E/libvqhip_t2w4.so $E/libvqhip_t3w8.so 2>&1 | grep -v amdgpu.ids | tee -a $O/shapes.txt
done
