for sh in "3072 16384 256 Cosine" "4096 16384 256 L2" "7311 3072 256 Cosine" "2048 8192 128 L2" "4096 8192 64 L2" "65536 16384 256 L2" "524288 16384 256 L2" "100352 8192 32 Cosine"; do
python tools/exp_shape.py $sh shipped build/exp/libvqhip_mid1.so build/exp/libvqhip_mid2.so 2>&1 | grep -v "Warn\|amdgpu"
done
