#!/bin/bash
# exact_stream_kernel under GPU sharing: P copies of the micro harness at once (each compares the keys of the streamed form,
# accumulated by atomicMin over all its launches, with the register form's).  usage: stream_stress.sh <binary> <P> <rounds> [args...]
bin=$1; P=$2; R=$3; shift 3
bad=0
for r in $(seq 1 $R); do
  for p in $(seq 1 $P); do $bin "$@" > gpurun_out/stress_$p.txt 2>&1 & done
  wait
  for p in $(seq 1 $P); do grep -q "mismatches 0$" gpurun_out/stress_$p.txt || { bad=$((bad+1)); grep -h "mismatches" gpurun_out/stress_$p.txt | cut -c1-200; }; done
done
echo "$bin $*: $bad of $((P*R)) processes with mismatches"
