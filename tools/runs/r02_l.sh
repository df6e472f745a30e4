#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_l
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r02_l/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 gpurun_out/r02_l/pytest.log | cut -c1-300
timeout 600 python bench.py --workload cvq --no-cpu-baseline > gpurun_out/r02_l/cvq.json 2> gpurun_out/r02_l/cvq.err; echo "cvq rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r02_l/cvq.json'))
print('cvq eager ms', d['ms_per_step'], 'graphed', d.get('module_graphed'))
PY
tail -3 gpurun_out/r02_l/cvq.err
