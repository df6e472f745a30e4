#!/bin/bash
# Round-2 evidence, one GPU box: every artefact lands under gpurun_out/r02_p/ and is copied into profiles/ afterwards.
cd /root/repo
P=gpurun_out/r02_p
mkdir -p $P
timeout 1800 python -m pytest tests -m gpu -q > $P/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $P/pytest.log | cut -c1-200
# 1. the driver's command
timeout 900 python bench.py > $P/bench.json 2> $P/bench.err; echo "bench rc=$?"
# 2. ONE process: bench.py JSON + rocprofv3 per-kernel summary (HIP-event time vs profiler average of the proposal kernel)
tools/prof_bench.sh r02_p/bench_prof --no-cpu-baseline
# 3. PMC passes of the same command (separate --pmc runs), summary -> pmc_latest.json
tools/pmc_passes.sh r02_p/pmc > $P/pmc_passes.log 2>&1; tail -3 $P/pmc_passes.log | cut -c1-400
# 4. BASELINE configs[2] (VQ-KD K=8192 D=32, 512 images x 196 tokens): kernel stats + PMC
tools/prof_shape.sh r02_p/c3_stats 100352 8192 32 Cosine
tools/pmc_shape.sh r02_p/c3_pmc 100352 8192 32 Cosine
# 5. the other workloads of bench.py and the shapes table
timeout 600 python bench.py --workload cvq --no-cpu-baseline > $P/cvq.json 2> $P/cvq.err; echo "cvq rc=$?"
timeout 600 python bench.py --workload cvq --images 256 --no-cpu-baseline > $P/cvq256.json 2> $P/cvq256.err; echo "cvq256 rc=$?"
timeout 600 python bench.py --workload tokenize --no-cpu-baseline > $P/tokenize.json 2> $P/tokenize.err; echo "tok rc=$?"
timeout 600 python bench.py --images 32 --no-cpu-baseline > $P/bench_32img.json 2> $P/bench_32img.err; echo "32img rc=$?"
timeout 600 python bench.py --images 256 --no-cpu-baseline > $P/bench_256img.json 2> $P/bench_256img.err; echo "256img rc=$?"
VQ_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --images 256 --steps 20 --warmup 5 > $P/gpus2_shared.json 2> $P/gpus2_shared.err; echo "gpus2 rc=$?"
VQ_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --workload cvq --steps 20 --warmup 5 > $P/gpus2_cvq_shared.json 2> $P/gpus2_cvq_shared.err; echo "gpus2 cvq rc=$?"
timeout 600 python tools/bench_shapes.py > $P/shapes.txt 2>&1; echo "shapes rc=$?"
timeout 600 python tools/bench_cvq.py 3072 > $P/training_steps.txt 2>&1
timeout 600 python tools/ab_filter.py > $P/ab_filter.txt 2>&1
for f in bench cvq cvq256 tokenize bench_32img bench_256img gpus2_shared gpus2_cvq_shared; do python3 - $P/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    extra = {k: (round(v['ms_per_step'], 4) if isinstance(v, dict) and 'ms_per_step' in v else v) for k, v in d.items() if k in ('ops_step', 'module_train', 'module_graphed', 'codebook_in_sync', 'rccl_ranks')}
    print(f"{sys.argv[1].split('/')[-1]:26s} value {d['value']/1e6:9.2f} M tok/s  ms/step {d['ms_per_step']:.4f}  frac {d['roofline']['frac']:.3f}  n_gpus {d['n_gpus']}  {extra}  cpu {d.get('cpu_baseline', {}).get('value')}")
except Exception as e:
    print(sys.argv[1], 'unreadable', e)
PY
done
