for sh in "3072 16384 256 Cosine" "524288 16384 256 L2"; do
python tools/exp_shape.py $sh shipped build/exp/libvqhip_bare.so build/exp/libvqhip_nodmaldsepi.so 2>&1 | grep -v "Warn\|amdgpu"
done
