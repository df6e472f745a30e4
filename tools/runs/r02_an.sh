#!/bin/bash
cd /root/repo
O=gpurun_out/r02_an; mkdir -p $O
E=build/exp
for shape in "3072 16384 256 Cosine" "4096 16384 256 L2" "2048 16384 256 L2" "16384 3072 256 Cosine"; do
timeout 600 python tools/exp_shape.py $shape shipped $E/libvqhip_small4k.so $E/libvqhip_small2k.so 2>&1 | grep -v amdgpu.ids | tee -a $O/shapes2.txt
done
