export VQHIP_LIB=build/exp/libvqhip_stage.so
for sh in "3072 16384 256 Cosine fp32" "65536 16384 256 L2 bf16"; do
  python tools/stage_stamps.py $sh 2>&1 | grep -v "amdgpu.ids\|Warning"
done
