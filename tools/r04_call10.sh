cd /root/repo
o=gpurun_out
timeout 2000 python -m pytest tests -x -q -m gpu > $o/r04_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $o/r04_gpu_tests.log | cut -c1-300
VQ_FUZZ_DIMS=8,16,32 timeout 500 python tools/fuzz_vs_exact.py 300 47 > $o/r04_fuzz_small_d.txt 2>&1; tail -2 $o/r04_fuzz_small_d.txt
timeout 300 python tools/fuzz_vs_exact.py 120 48 > $o/r04_fuzz_all_d.txt 2>&1; tail -2 $o/r04_fuzz_all_d.txt
python tools/bench_shapes.py > $o/r04_shapes.txt 2>&1; cat $o/r04_shapes.txt | head -60
