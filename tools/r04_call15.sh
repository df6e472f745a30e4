cd /tmp && export TMPDIR=/tmp
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 TORCHELASTIC_RUN_ID=manual VQ_FORCE_EXCHANGE=1
for route in direct torch; do
  export VQHIP_ALLREDUCE=$route MASTER_PORT=$((29600 + RANDOM % 100))
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/r04_rccl_trace_$route -- python3 /root/repo/bench.py --workload cvq --steps 30 --warmup 5 --min-seconds 0 --no-cpu-baseline > /root/repo/gpurun_out/r04_rccl_trace_$route.json 2> /root/repo/gpurun_out/r04_rccl_trace_$route.err
  echo "$route rc=$?"
  python3 /root/repo/tools/step_timeline.py /root/repo/gpurun_out/r04_rccl_trace_$route 'pre_kernel<0, true' 3 > /root/repo/gpurun_out/r04_rccl_timeline_$route.txt 2>&1
  echo "== $route (graphed replay, third step from the end)"; cat /root/repo/gpurun_out/r04_rccl_timeline_$route.txt | cut -c1-110
done
