#!/bin/bash
# group records + replay identification in the filtered proposal kernels: parity, fuzz, A/B against the per-element form
cd /root/repo
O=gpurun_out/r02_z; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-200
VQ_FUZZ_DIMS=8,16,32,32,64,128 timeout 400 python tools/fuzz_vs_exact.py 200 11 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log
E=build/exp
for shape in "100352 8192 32 Cosine" "65536 8192 32 Cosine" "524288 16384 8 L2" "65536 8192 64 L2" "65536 8192 128 Cosine" "3072 8192 32 Cosine"; do
timeout 600 python tools/exp_shape.py $shape shipped $E/libvqhip_elem.so $E/libvqhip_nobranch.so 2>&1 | grep -v amdgpu.ids | tee -a $O/shapes.txt
done
