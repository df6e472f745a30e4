#!/bin/bash
# vqhip_encode (fused front): GPU suite, fuzz (col pass uses the fused front), benches
cd /root/repo
O=gpurun_out/r02_aq; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
timeout 400 python tools/fuzz_vs_exact.py 150 53 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log | cut -c1-300
timeout 600 python bench.py --workload cvq --no-cpu-baseline > $O/cvq.json 2> $O/cvq.err; echo "cvq rc=$?"
timeout 600 python bench.py --images 32 --no-cpu-baseline > $O/b32.json 2> $O/b32.err
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b.json 2> $O/b.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r02_aq/cvq.json').read().strip().splitlines()[-1])
print('cvq eager ms', d['ms_per_step'], 'graphed', d['module_graphed']['ms_per_step'])
d = json.loads(open('gpurun_out/r02_aq/b32.json').read().strip().splitlines()[-1])
print('32 images:', d['value'] / 1e6, 'M tok/s', d['ms_per_step'], 'ms; ops', d['ops_step']['ms_per_step'])
d = json.loads(open('gpurun_out/r02_aq/b.json').read().strip().splitlines()[-1])
print('headline:', d['value'] / 1e6, 'M tok/s', d['ms_per_step'], 'ms; kernel', d['roofline']['kernel_ms'], 'ops', d['ops_step']['ms_per_step'])
PY
