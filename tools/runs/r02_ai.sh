#!/bin/bash
cd /root/repo
O=gpurun_out/r02_ai; mkdir -p $O
VQHIP_LIB=build/exp/libvqhip_stamps.so timeout 120 python tools/stamps_run.py 100352 8192 32 Cosine 2>&1 | grep stamps | sort | head -40 | tee $O/stamps_c3.txt
VQHIP_LIB=build/exp/libvqhip_stamps.so timeout 120 python tools/stamps_run.py 65536 8192 32 Cosine 2>&1 | grep stamps | sort | head -12 | tee $O/stamps_64k.txt
