#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_j
for lib in shipped max1 max2; do
  if [ $lib = shipped ]; then unset VQHIP_LIB; else export VQHIP_LIB=/root/repo/build/exp/libvqhip_$lib.so; fi
  VQ_FUZZ_DIMS=64,128 timeout 300 python tools/fuzz_vs_exact.py 70 7 > gpurun_out/r02_j/fuzz_$lib.txt 2>&1; echo "$lib rc=$?"; tail -3 gpurun_out/r02_j/fuzz_$lib.txt
done
