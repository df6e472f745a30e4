cd /root/repo
o=gpurun_out
python tools/exp_shape.py 524288 16384 8 Cosine shipped build/exp/libvqhip_noatomic.so build/exp/libvqhip_nofrag.so build/exp/libvqhip_noboth.so > $o/r04_epilogue_shares.txt 2>&1
python tools/exp_shape.py 100352 8192 32 Cosine shipped build/exp/libvqhip_noatomic.so build/exp/libvqhip_nofrag.so build/exp/libvqhip_noboth.so >> $o/r04_epilogue_shares.txt 2>&1
cat $o/r04_epilogue_shares.txt
