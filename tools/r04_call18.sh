cd /root/repo
o=gpurun_out
timeout 2000 python -m pytest tests -x -q -m gpu > $o/r04_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $o/r04_gpu_tests.log | cut -c1-200
export VQ_EXP_ENCODE=1
python tools/exp_shape.py 100352 8192 32 Cosine build/exp/libvqhip_prev.so shipped build/exp/libvqhip_dec256.so build/exp/libvqhip_dec512.so > $o/r04_cos_front.txt 2>&1
python tools/exp_shape.py 3072 16384 256 Cosine build/exp/libvqhip_prev.so shipped build/exp/libvqhip_dec256.so >> $o/r04_cos_front.txt 2>&1
python tools/exp_shape.py 524288 16384 8 Cosine build/exp/libvqhip_prev.so shipped build/exp/libvqhip_dec256.so build/exp/libvqhip_dec512.so >> $o/r04_cos_front.txt 2>&1
unset VQ_EXP_ENCODE
python tools/exp_shape.py 524288 16384 256 L2 shipped build/exp/libvqhip_dec256.so build/exp/libvqhip_dec512.so >> $o/r04_cos_front.txt 2>&1
cat $o/r04_cos_front.txt
