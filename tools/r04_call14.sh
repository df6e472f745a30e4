cd /root/repo
o=gpurun_out
timeout 2000 python -m pytest tests -x -q -m gpu > $o/r04_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $o/r04_gpu_tests.log | cut -c1-300
VQ_FUZZ_DIMS=8,16,24,32 timeout 300 python tools/fuzz_vs_exact.py 150 53 > $o/r04_fuzz_small_d.txt 2>&1; tail -2 $o/r04_fuzz_small_d.txt
VQ_PROF_ENCODE=1 bash tools/timeline_shape.sh r04_c3 pre_kernel 100352 8192 32 Cosine
VQ_PROF_ENCODE=1 VQ_PROF_BF16=1 bash tools/timeline_shape.sh r04_tok pre_kernel 524288 16384 8 L2
python tools/bench_shapes.py 2>&1 | grep "C3\|C5"
python bench.py --workload tokenize --no-cpu-baseline --min-seconds 3 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tokenize', d['value']/1e6, d['ms_per_step'], d['parity']['mismatches'])"
