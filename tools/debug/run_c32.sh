for sh in "100352 8192 32 Cosine" "131072 8192 32 Cosine" "524288 16384 8 L2"; do
python tools/exp_shape.py $sh shipped build/exp/libvqhip_c32nodma.so build/exp/libvqhip_c32nocomp.so 2>&1 | grep -v "Warn\|amdgpu"
done
