#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_e
E=build/exp
timeout 1200 python tools/exp_shape.py 100352 8192 32 Cosine shipped $E/libvqhip_tps8.so $E/libvqhip_s1.so $E/libvqhip_tps8_s1.so $E/libvqhip_s1_tt2.so $E/libvqhip_tps8_s1_tt2.so 2>&1 | tee gpurun_out/r02_e/c3.txt
timeout 1200 python tools/exp_shape.py 524288 16384 8 L2 shipped $E/libvqhip_tps8.so $E/libvqhip_s1.so $E/libvqhip_tps8_s1.so $E/libvqhip_s1_tt2.so $E/libvqhip_tps8_s1_tt2.so 2>&1 | tee gpurun_out/r02_e/d8.txt
timeout 1200 python tools/exp_shape.py 12544 8192 32 Cosine shipped $E/libvqhip_tps8.so $E/libvqhip_s1.so $E/libvqhip_tps8_s1.so 2>&1 | tee gpurun_out/r02_e/c3small.txt
