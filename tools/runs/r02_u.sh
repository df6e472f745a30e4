#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_u
timeout 1000 python tools/fuzz_vs_exact.py 420 101 > gpurun_out/r02_u/fuzz_a.txt 2>&1; echo "fuzz a rc=$?"; tail -3 gpurun_out/r02_u/fuzz_a.txt
VQ_FUZZ_DIMS=8,16,24,32,64,128 timeout 1000 python tools/fuzz_vs_exact.py 420 102 > gpurun_out/r02_u/fuzz_b.txt 2>&1; echo "fuzz b rc=$?"; tail -3 gpurun_out/r02_u/fuzz_b.txt
