#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r02_w
timeout 1800 python -m pytest tests -m gpu -q -x > gpurun_out/r02_w/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r02_w/pytest.log | cut -c1-200
for shape in "524288 16384 8 L2" "262144 8192 32 Cosine" "100352 8192 32 Cosine"; do
  timeout 900 python tools/exp_shape.py $shape shipped 2>&1 | tail -1 | sed "s/^/$shape: /"
done | tee gpurun_out/r02_w/check.txt
timeout 600 python bench.py --workload tokenize --no-cpu-baseline > gpurun_out/r02_w/tokenize.json 2>/dev/null; python3 -c "
import json; d=json.load(open('gpurun_out/r02_w/tokenize.json')); print('tokenize', d['value']/1e6, 'M tok/s', d['ms_per_step'])"
