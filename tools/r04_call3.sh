cd /root/repo
o=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > $o/r04_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -15 $o/r04_gpu_tests.log
python tools/exp_shape.py 100352 8192 32 Cosine build/exp/libvqhip_before.so shipped > $o/r04_identify_ab.txt 2>&1
python tools/exp_shape.py 524288 16384 8 Cosine build/exp/libvqhip_before.so shipped >> $o/r04_identify_ab.txt 2>&1
python tools/exp_shape.py 65536 8192 32 Cosine build/exp/libvqhip_before.so shipped >> $o/r04_identify_ab.txt 2>&1
python tools/exp_shape.py 20000 8192 32 Cosine build/exp/libvqhip_before.so shipped >> $o/r04_identify_ab.txt 2>&1
cat $o/r04_identify_ab.txt
VQ_FUZZ_DIMS=8,16,32 timeout 400 python tools/fuzz_vs_exact.py 240 41 > $o/r04_fuzz_small_d.txt 2>&1; tail -3 $o/r04_fuzz_small_d.txt
