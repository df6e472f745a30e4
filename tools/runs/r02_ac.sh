#!/bin/bash
# after NOAUX + group records: full GPU suite, fuzz, configs[2] kernel stats + PMC
cd /root/repo
O=gpurun_out/r02_ac; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
timeout 400 python tools/fuzz_vs_exact.py 200 23 > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log | cut -c1-300
tools/prof_shape.sh r02_ac/c3_stats 100352 8192 32 Cosine
tools/pmc_shape.sh r02_ac/c3_pmc 100352 8192 32 Cosine
timeout 600 python bench.py --workload tokenize --no-cpu-baseline > $O/tokenize.json 2> $O/tokenize.err; echo "tok rc=$?"; cut -c1-400 $O/tokenize.json
