cd /root/repo
o=gpurun_out
timeout 2000 python -m pytest tests -x -q -m gpu > $o/r04_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -4 $o/r04_gpu_tests.log | cut -c1-300
timeout 300 python tools/fuzz_vs_exact.py 150 57 > $o/r04_fuzz_all_d.txt 2>&1; tail -2 $o/r04_fuzz_all_d.txt
VQ_PROF_ENCODE=1 bash tools/timeline_shape.sh r04_c3 pre_kernel 100352 8192 32 Cosine
VQ_PROF_ENCODE=1 bash tools/timeline_shape.sh r04_c4 pre_kernel 3072 16384 256 Cosine
python tools/bench_shapes.py 2>&1 | grep "C3\|C4\|C5\|cluster"
python bench.py --workload cvq --no-cpu-baseline --min-seconds 3 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['cvq']; print('cvq', c['ms_per_step'], c.get('ms_per_step_graphed'))"
