#!/bin/bash
# The shared-GPU rehearsal of `bench.py --gpus 8` (tests/test_gpu_two_ranks.py) in a loop: the stderr of a failing run.
# usage: tools/debug/eight_ranks_loop.sh <runs> [extra env, e.g. VQ_TUNE18=0]
n=${1:-4}
for i in $(seq 1 $n); do
  VQ_BENCH_SHARE_GPU=1 VQ_BENCH_CVQ_SETTLE=30 python bench.py --gpus 8 --steps 3 --warmup 1 --images 16 --min-seconds 0 --no-cpu-baseline > gpurun_out/e8_$i.out 2> gpurun_out/e8_$i.err
  echo "run $i rc=$?"
  grep -h "parity self-check FAILED\|Traceback\|Error" gpurun_out/e8_$i.err | cut -c1-1200 | head -5
done
