cd /root/repo
o=gpurun_out
export VQ_EXP_ENCODE=1
python tools/exp_shape.py 100352 8192 32 Cosine build/exp/libvqhip_before.so build/exp/libvqhip_prev.so shipped > $o/r04_cos_front.txt 2>&1
python tools/exp_shape.py 3072 16384 256 Cosine build/exp/libvqhip_before.so build/exp/libvqhip_prev.so shipped >> $o/r04_cos_front.txt 2>&1
python tools/exp_shape.py 65536 8192 768 Cosine build/exp/libvqhip_prev.so shipped >> $o/r04_cos_front.txt 2>&1
cat $o/r04_cos_front.txt
