#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_t
E=build/exp
timeout 900 python tools/exp_shape.py 65536 8192 1024 L2 shipped $E/libvqhip_d1024_w4t2.so 2>&1 | tee gpurun_out/r02_t/d1024.txt
timeout 900 python tools/exp_shape.py 65536 8192 768 Cosine shipped $E/libvqhip_d768_w4t4.so 2>&1 | tee gpurun_out/r02_t/d768.txt
timeout 900 python tools/exp_shape.py 8192 8192 768 Cosine shipped $E/libvqhip_d768_w4t4.so 2>&1 | tee gpurun_out/r02_t/d768s.txt
timeout 900 python tools/exp_shape.py 65536 16384 512 L2 shipped $E/libvqhip_d512_w4t4.so 2>&1 | tee gpurun_out/r02_t/d512.txt
