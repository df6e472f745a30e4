cd /root/repo
o=gpurun_out
timeout 900 python -m pytest tests/test_gpu_rccl.py -q > $o/r04_rccl_test.log 2>&1; echo "rccl test rc=$?"; tail -3 $o/r04_rccl_test.log
timeout 900 python -m pytest tests/test_gpu_modules.py -q -k "full_size" > $o/r04_fullsize_test.log 2>&1; echo "fullsize rc=$?"; tail -5 $o/r04_fullsize_test.log
python tools/exp_shape.py 100352 8192 32 Cosine shipped shipped@6=0 build/exp/libvqhip_noreplay.so build/exp/libvqhip_noreplay.so@6=0 > $o/r04_c3_shares.txt 2>&1
python tools/exp_shape.py 524288 16384 8 Cosine shipped shipped@6=0 build/exp/libvqhip_noreplay.so >> $o/r04_c3_shares.txt 2>&1
python tools/exp_shape.py 100352 8192 32 L2 shipped shipped@6=0 >> $o/r04_c3_shares.txt 2>&1
cat $o/r04_c3_shares.txt
