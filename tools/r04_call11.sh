cd /root/repo
o=gpurun_out
timeout 2000 python -m pytest tests -x -q -m gpu > $o/r04_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $o/r04_gpu_tests.log | cut -c1-300
python tools/exp_shape.py 100352 8192 32 Cosine build/exp/libvqhip_before.so build/exp/libvqhip_gt4_top1.so shipped > $o/r04_top2.txt 2>&1
python tools/exp_shape.py 524288 16384 8 Cosine build/exp/libvqhip_before.so build/exp/libvqhip_gt4_top1.so shipped >> $o/r04_top2.txt 2>&1
python tools/exp_shape.py 20000 8192 32 Cosine build/exp/libvqhip_before.so build/exp/libvqhip_gt4_top1.so shipped >> $o/r04_top2.txt 2>&1
cat $o/r04_top2.txt
VQ_FUZZ_DIMS=8,16,32 timeout 400 python tools/fuzz_vs_exact.py 200 51 > $o/r04_fuzz_small_d.txt 2>&1; tail -2 $o/r04_fuzz_small_d.txt
python tools/bench_shapes.py 2>&1 | grep "C3\|C5" 
bash tools/prof_shape.sh r04_tok_prof_e 524288 16384 8 Cosine
