cd /root/repo
o=gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > $o/r04_gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $o/r04_gpu_tests.log | cut -c1-200
python tools/exp_shape.py 100352 8192 32 Cosine build/exp/libvqhip_before.so build/exp/libvqhip_gt4.so shipped > $o/r04_group_tiles.txt 2>&1
python tools/exp_shape.py 524288 16384 8 Cosine build/exp/libvqhip_before.so build/exp/libvqhip_gt4.so shipped >> $o/r04_group_tiles.txt 2>&1
python tools/exp_shape.py 20000 8192 32 Cosine build/exp/libvqhip_before.so build/exp/libvqhip_gt4.so shipped >> $o/r04_group_tiles.txt 2>&1
python tools/exp_shape.py 65536 16384 8 L2 build/exp/libvqhip_before.so build/exp/libvqhip_gt4.so shipped >> $o/r04_group_tiles.txt 2>&1
cat $o/r04_group_tiles.txt
bash tools/prof_shape.sh r04_c3_prof_d 100352 8192 32 Cosine
bash tools/prof_shape.sh r04_tok_prof_d 524288 16384 8 Cosine
