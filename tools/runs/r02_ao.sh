#!/bin/bash
# kernel timeline of the eval module step at 32 images (8192 tokens) and GPU tests with the new D=256 threshold
cd /root/repo
O=gpurun_out/r02_ao; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/b32 -- python3 /root/repo/bench.py --images 32 --no-cpu-baseline --steps 30 --warmup 5 > /root/repo/$O/b32.json 2> /root/repo/$O/b32.err
cd /root/repo
python3 tools/timeline.py $O/b32 cb_stats_kernel 20 > $O/timeline.txt 2>&1; cut -c1-110 $O/timeline.txt
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r02_ao/b32.json').read().strip().splitlines()[-1])
print('32 images:', d['value'] / 1e6, 'M tok/s', d['ms_per_step'], 'ms; ops', d['ops_step']['ms_per_step'])
PY
