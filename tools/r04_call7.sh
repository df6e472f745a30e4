cd /root/repo
o=gpurun_out
python tools/exp_shape.py 100352 8192 32 Cosine shipped shipped@2=2 shipped@2=1 > $o/r04_slices.txt 2>&1
python tools/exp_shape.py 65536 8192 32 Cosine shipped shipped@2=2 >> $o/r04_slices.txt 2>&1
python tools/exp_shape.py 200000 8192 32 Cosine shipped shipped@2=2 >> $o/r04_slices.txt 2>&1
python tools/exp_shape.py 524288 16384 8 Cosine shipped shipped@2=2 >> $o/r04_slices.txt 2>&1
cat $o/r04_slices.txt
