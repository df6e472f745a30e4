#!/bin/bash
cd /root/repo
O=gpurun_out/r02_ap; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
timeout 600 python bench.py --workload cvq --no-cpu-baseline > $O/cvq.json 2> $O/cvq.err; echo "cvq rc=$?"
timeout 600 python bench.py --images 32 --no-cpu-baseline > $O/b32.json 2> $O/b32.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r02_ap/cvq.json').read().strip().splitlines()[-1])
print('cvq eager ms', d['ms_per_step'], 'graphed', d['module_graphed']['ms_per_step'])
d = json.loads(open('gpurun_out/r02_ap/b32.json').read().strip().splitlines()[-1])
print('32 images:', d['value'] / 1e6, 'M tok/s', d['ms_per_step'], 'ms; ops', d['ops_step']['ms_per_step'])
PY
