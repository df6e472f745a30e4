export VQ_TRAIN_NO_GRAPH=1
echo "== default"; python tools/bench_train_shapes.py cvq 2>&1 | grep -v "Warn\|amdgpu\|return Var"
echo "== pool 128"; VQ_TRAIN_POOL=128 python tools/bench_train_shapes.py cvq 2>&1 | grep -v "Warn\|amdgpu\|return Var"
echo "== hook by hook"; VQHIP_ONE_CALL=0 python tools/bench_train_shapes.py cvq 2>&1 | grep -v "Warn\|amdgpu\|return Var"
echo "== bench"; python bench.py --workload cvq --no-cpu-baseline 2>&1 | grep -v "Warn\|amdgpu\|return Var" | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print({k:r['cvq'].get(k) for k in ('ms_per_step','ms_per_step_graphed','exchange_rows','one_call_forward')})
"
