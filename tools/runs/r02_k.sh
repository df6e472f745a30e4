#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_k
E=build/exp
for lib in shipped tm2 tm4 tm5; do
  if [ $lib = shipped ]; then unset VQHIP_LIB; else export VQHIP_LIB=/root/repo/$E/libvqhip_$lib.so; fi
  VQ_FUZZ_DIMS=64,128,32 timeout 300 python tools/fuzz_vs_exact.py 80 7 > gpurun_out/r02_k/fuzz_$lib.txt 2>&1; echo "$lib rc=$?"; tail -2 gpurun_out/r02_k/fuzz_$lib.txt
done
unset VQHIP_LIB
timeout 900 python tools/exp_shape.py 100352 8192 32 Cosine shipped $E/libvqhip_tm2.so $E/libvqhip_tm4.so $E/libvqhip_tm5.so 2>&1 | tee gpurun_out/r02_k/c3.txt
timeout 900 python tools/exp_shape.py 65536 16384 128 L2 shipped $E/libvqhip_tm2.so $E/libvqhip_tm4.so $E/libvqhip_tm5.so 2>&1 | tee gpurun_out/r02_k/d128.txt
timeout 900 python tools/exp_shape.py 524288 16384 8 L2 shipped $E/libvqhip_tm2.so $E/libvqhip_tm4.so $E/libvqhip_tm5.so 2>&1 | tee gpurun_out/r02_k/d8.txt
