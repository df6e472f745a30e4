#!/bin/bash
# bench.py contract checks on one GPU: default line, cvq / tokenize workloads, bare --gpus 2 self-launch (shared GPU, gloo)
cd /root/repo
mkdir -p gpurun_out/r02_b
timeout 900 python bench.py > gpurun_out/r02_b/bench.json 2> gpurun_out/r02_b/bench.err; echo "bench rc=$?"
timeout 600 python bench.py --workload cvq --no-cpu-baseline > gpurun_out/r02_b/cvq.json 2> gpurun_out/r02_b/cvq.err; echo "cvq rc=$?"
timeout 600 python bench.py --workload cvq --images 256 --no-cpu-baseline > gpurun_out/r02_b/cvq256.json 2> gpurun_out/r02_b/cvq256.err; echo "cvq256 rc=$?"
timeout 600 python bench.py --workload tokenize --no-cpu-baseline > gpurun_out/r02_b/tok.json 2> gpurun_out/r02_b/tok.err; echo "tok rc=$?"
VQ_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --images 256 > gpurun_out/r02_b/g2.json 2> gpurun_out/r02_b/g2.err; echo "gpus2 rc=$?"
VQ_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --steps 10 --warmup 3 --workload cvq > gpurun_out/r02_b/g2cvq.json 2> gpurun_out/r02_b/g2cvq.err; echo "gpus2 cvq rc=$?"
for f in bench cvq cvq256 tok g2 g2cvq; do echo "== $f"; head -c 1500 gpurun_out/r02_b/$f.json; echo; tail -3 gpurun_out/r02_b/$f.err; done
