for sh in "65536 8192 128 L2" "65536 16384 64 L2" "8192 8192 128 L2" "262144 8192 128 Cosine" "65536 16384 256 L2"; do
python tools/exp_shape.py $sh shipped build/exp/libvqhip_sbf1.so build/exp/libvqhip_sbf2.so build/exp/libvqhip_sbf3.so 2>&1 | grep -v "Warn\|amdgpu"
done
