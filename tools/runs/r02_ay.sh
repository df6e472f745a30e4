#!/bin/bash
cd /root/repo
E=build/exp
for shape in "100352 8192 32 Cosine" "65536 8192 32 Cosine" "524288 16384 8 L2"; do
timeout 600 python tools/exp_shape.py $shape shipped $E/libvqhip_dma2.so $E/libvqhip_dma4.so 2>&1 | grep -v amdgpu.ids
done
