#!/bin/bash
# group records (fixed): parity, fuzz, A/B key 9, branch variants; grid-barrier micro-benchmark
cd /root/repo
O=gpurun_out/r02_aa; mkdir -p $O
timeout 120 ./build/gridbar 2>&1 | tee $O/gridbar.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-200
VQ_FUZZ_DIMS=8,16,32,32,64,128 timeout 400 python tools/fuzz_vs_exact.py 240 11 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz.log | cut -c1-300
timeout 600 python tools/ab_key.py 9 2>&1 | grep -v amdgpu.ids | tee $O/ab_groups.txt
E=build/exp
for shape in "100352 8192 32 Cosine" "65536 8192 32 Cosine" "524288 16384 8 L2" "16384 8192 32 Cosine"; do
timeout 600 python tools/exp_shape.py $shape shipped $E/libvqhip_br2.so $E/libvqhip_br99.so 2>&1 | grep -v amdgpu.ids | tee -a $O/shapes.txt
done
