for sh in "524288 16384 256 L2" "65536 16384 256 L2" "8192 16384 256 L2"; do
python tools/exp_shape.py $sh shipped build/exp/libvqhip_sb2pf2.so build/exp/libvqhip_sb2pf3.so build/exp/libvqhip_sb2pf4.so 2>&1 | grep -v "Warn\|amdgpu"
done
