#!/bin/bash
# headline evidence only: the driver's command, the same command under rocprofv3 (one process, both artefacts), PMC passes
cd /root/repo
P=gpurun_out/r02_h
mkdir -p $P
timeout 900 python bench.py > $P/bench.json 2> $P/bench.err; echo "bench rc=$?"
tools/prof_bench.sh r02_h/bench_prof --no-cpu-baseline
tools/pmc_passes.sh r02_h/pmc > $P/pmc_passes.log 2>&1; tail -2 $P/pmc_passes.log | cut -c1-300
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r02_h/bench.json').read().strip().splitlines()[-1])
print('bench:', d['value'] / 1e6, d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['cpu_baseline']['value'])
PY
