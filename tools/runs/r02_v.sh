#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_v
E=build/exp
for shape in "100352 8192 32 Cosine" "65536 8192 32 Cosine" "196608 8192 32 Cosine" "524288 16384 8 L2"; do
  timeout 900 python tools/exp_shape.py $shape shipped $E/libvqhip_nbuf2.so $E/libvqhip_nbuf3.so 2>&1 | tail -4
done | tee gpurun_out/r02_v/ring.txt
