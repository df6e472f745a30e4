#!/bin/bash
# GPU suite + CVQ/VQGAN bench lines + fuzz + one graphed CVQ step as a kernel timeline
cd /root/repo
O=gpurun_out/r02_x; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"
tail -25 $O/pytest.log | cut -c1-300
timeout 600 python bench.py --workload cvq --no-cpu-baseline > $O/cvq.json 2> $O/cvq.err; echo "cvq rc=$?"
timeout 600 python bench.py --workload vqgan --no-cpu-baseline > $O/vqgan.json 2> $O/vqgan.err; echo "vqgan rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r02_x/cvq.json'))
print('cvq eager ms', d['ms_per_step'], 'graphed', d.get('module_graphed'))
d = json.load(open('gpurun_out/r02_x/vqgan.json'))
print('vqgan', d['value'], d['ms_per_step'], d.get('ops_step'), d.get('module_train'))
PY
timeout 400 python tools/fuzz_vs_exact.py 240 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/cvq -- python3 /root/repo/bench.py --workload cvq --no-cpu-baseline --steps 20 --warmup 5 > /root/repo/$O/cvq_prof.json 2> /root/repo/$O/cvq_prof.err
cd /root/repo
python3 tools/timeline.py $O/cvq vq_backward_kernel 3 > $O/timeline.txt 2>&1; cat $O/timeline.txt | cut -c1-110
