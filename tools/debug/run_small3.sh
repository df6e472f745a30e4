for sh in "3072 16384 256 Cosine" "4096 16384 256 L2" "7311 3072 256 Cosine" "8192 16384 256 L2" "2048 8192 128 L2" "6272 8192 768 Cosine"; do
python tools/exp_shape.py $sh shipped build/exp/libvqhip_lag48.so build/exp/libvqhip_nofence.so build/exp/libvqhip_small3.so 2>&1 | grep -v "Warn\|amdgpu"
done
