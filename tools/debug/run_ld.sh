for sh in "65536 8192 768 L2" "6272 8192 768 Cosine" "65536 8192 1024 L2" "65536 8192 512 L2"; do
python tools/exp_shape.py $sh shipped build/exp/libvqhip_ld1.so build/exp/libvqhip_ld2.so build/exp/libvqhip_ld3.so 2>&1 | grep -v "Warn\|amdgpu"
done
