#!/bin/bash
# round-2 first GPU pass: parity suite on the regenerated (reference-import) fixtures, baseline bench, C3 profile + PMC
cd /root/repo
mkdir -p gpurun_out/r02_a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_a/pytest.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/r02_a/pytest.log
timeout 600 python bench.py > gpurun_out/r02_a/bench.json 2> gpurun_out/r02_a/bench.err; echo "bench rc=$?"
timeout 600 python tools/bench_shapes.py > gpurun_out/r02_a/shapes.txt 2>&1; echo "shapes rc=$?"
timeout 600 tools/prof_shape.sh r02_a/c3_stats 100352 8192 32 Cosine
timeout 900 tools/pmc_shape.sh r02_a/c3_pmc 100352 8192 32 Cosine
timeout 600 tools/prof_shape.sh r02_a/c4_stats 3072 16384 256 Cosine
