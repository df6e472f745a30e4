#!/bin/bash
# A/B of the two routes of the packed all-reduce on a world-size-1 RCCL group (run on the GPU box): alternating processes.
# usage: tools/rccl_ws1_ab.sh <tag> [rounds]   -> gpurun_out/<tag>_{torch,direct}_<i>.json + a summary
tag=$1; rounds=${2:-3}
cd /root/repo
for i in $(seq 1 $rounds); do
  for route in torch direct; do
    VQ_FORCE_EXCHANGE=1 VQHIP_ALLREDUCE=$route timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port $((29520 + i)) bench.py --workload cvq --min-seconds 2 --no-cpu-baseline > gpurun_out/${tag}_${route}_$i.json 2> gpurun_out/${tag}_${route}_$i.err || echo "$route $i failed"
  done
done
python3 - $tag $rounds <<'PY'
import json, sys
tag, rounds = sys.argv[1], int(sys.argv[2])
for route in ('torch', 'direct'):
    rows = []
    for i in range(1, rounds + 1):
        try:
            d = json.loads(open(f'gpurun_out/{tag}_{route}_{i}.json').read().strip().splitlines()[-1])
        except Exception as e:
            print(route, i, 'ERR', e); continue
        c = d['cvq']
        rows.append((c['ms_per_step'], c.get('ms_per_step_graphed'), c['collective_ms'], c['exchange_route']['direct'], c.get('graphed_error')))
    print(route, [tuple(round(v, 4) if isinstance(v, float) else v for v in r) for r in rows])
PY
