for n in 32768 65536 98304 100352 131072; do
python tools/exp_shape.py $n 8192 32 Cosine shipped 2>&1 | grep -v "Warn\|amdgpu"
done
