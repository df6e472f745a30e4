for sh in "524288 16384 256 L2" "65536 16384 256 L2" "3072 16384 256 Cosine" "7311 3072 256 Cosine" "65536 8192 512 L2" "65536 8192 128 L2"; do
python tools/exp_shape.py $sh shipped build/exp/libvqhip_newdef2.so build/exp/libvqhip_spf1.so 2>&1 | grep -v "Warn\|amdgpu"
done
