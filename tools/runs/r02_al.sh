#!/bin/bash
# scalar wave index in the proposal kernel: parity + A/B against the vector form
cd /root/repo
O=gpurun_out/r02_al; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-300
timeout 300 python tools/fuzz_vs_exact.py 100 47 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log | cut -c1-300
E=build/exp
for shape in "100352 8192 32 Cosine" "65536 8192 32 Cosine" "524288 16384 8 L2" "524288 16384 256 L2" "8192 16384 256 L2" "3072 16384 256 Cosine" "65536 8192 128 Cosine" "65536 8192 1024 L2"; do
timeout 600 python tools/exp_shape.py $shape shipped $E/libvqhip_vwave.so 2>&1 | grep -v amdgpu.ids | tee -a $O/shapes.txt
done
