#!/bin/bash
cd /root/repo
O=gpurun_out/r02_z; mkdir -p $O
VQ_FUZZ_VERBOSE=1 VQ_FUZZ_DIMS=8,16,32,32,64,128 timeout 400 python tools/fuzz_vs_exact.py 200 11 > $O/fuzz2.log 2>&1; echo "fuzz rc=$?"; tail -4 $O/fuzz2.log
