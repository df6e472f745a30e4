#!/bin/bash
cd /root/repo
O=gpurun_out/r02_ab; mkdir -p $O
E=build/exp
for shape in "100352 8192 32 Cosine" "65536 8192 32 Cosine" "524288 16384 8 L2" "16384 8192 32 Cosine" "262144 8192 32 Cosine"; do
timeout 600 python tools/exp_shape.py $shape shipped $E/libvqhip_br2.so $E/libvqhip_rb2.so $E/libvqhip_rb6.so 2>&1 | grep -v amdgpu.ids | tee -a $O/shapes.txt
done
timeout 600 python tools/ab_key.py 9 2>&1 | grep -v amdgpu.ids | tee $O/ab_groups.txt
