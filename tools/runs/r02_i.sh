#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_i
VQ_FUZZ_DIMS=64,128 VQ_FILTER=1 timeout 300 python tools/fuzz_vs_exact.py 100 7 > gpurun_out/r02_i/fuzz_f1.txt 2>&1; echo "filter on rc=$?"; tail -12 gpurun_out/r02_i/fuzz_f1.txt
VQ_FUZZ_DIMS=64,128 VQ_FILTER=0 timeout 300 python tools/fuzz_vs_exact.py 100 7 > gpurun_out/r02_i/fuzz_f0.txt 2>&1; echo "filter off rc=$?"; tail -12 gpurun_out/r02_i/fuzz_f0.txt
VQ_FUZZ_DIMS=8,16,32 VQ_FILTER=1 timeout 300 python tools/fuzz_vs_exact.py 60 8 > gpurun_out/r02_i/fuzz_d32.txt 2>&1; echo "d32 rc=$?"; tail -6 gpurun_out/r02_i/fuzz_d32.txt
VQ_FUZZ_DIMS=256 timeout 300 python tools/fuzz_vs_exact.py 60 9 > gpurun_out/r02_i/fuzz_d256.txt 2>&1; echo "d256 rc=$?"; tail -6 gpurun_out/r02_i/fuzz_d256.txt
