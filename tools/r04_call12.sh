cd /root/repo
o=gpurun_out
python bench.py > $o/r04a_bench.json 2> $o/r04a_bench.err; echo "bench rc=$?"
python bench.py --workload tokenize --no-cpu-baseline > $o/r04a_tokenize.json 2>> $o/r04a_bench.err; echo "tokenize rc=$?"
python bench.py --workload cvq --no-cpu-baseline --min-seconds 3 > $o/r04a_cvq.json 2>> $o/r04a_bench.err; echo "cvq rc=$?"
for route in torch direct; do
  VQ_FORCE_EXCHANGE=1 VQHIP_ALLREDUCE=$route timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29511 bench.py --workload cvq --min-seconds 3 --no-cpu-baseline > $o/r04a_rccl_ws1_$route.json 2> $o/r04a_rccl_ws1_$route.err; echo "bench $route rc=$?"
done
python - <<'PY'
import json
for f in ['r04a_bench','r04a_tokenize','r04a_cvq','r04a_rccl_ws1_torch','r04a_rccl_ws1_direct']:
    try:
        d=json.loads(open('gpurun_out/'+f+'.json').read().strip().splitlines()[-1])
    except Exception as e:
        print(f,'ERR',e); continue
    c=d.get('cvq') or {}
    print(f, 'value %.1f M' % (d['value']/1e6), 'ms', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), 'kern_ms', round(d['roofline']['kernel_ms'],4), 'blocks', d['repeats']['blocks'],
          'graphed', c.get('ms_per_step_graphed'), 'coll_ms', c.get('collective_ms'), (c.get('exchange_route') or {}).get('direct'), d.get('parity',{}).get('mismatches'))
PY
