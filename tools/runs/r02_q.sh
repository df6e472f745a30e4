#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_q
for N in 65536 100352 131072 196608 262144; do
  timeout 300 python tools/exp_shape.py $N 8192 32 Cosine shipped 2>&1 | tail -1 | sed "s/^/N=$N /"
done | tee gpurun_out/r02_q/balance.txt
