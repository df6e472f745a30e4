export VQHIP_LIB=build/exp/libvqhip_phase.so
for sh in "3072 16384 256 Cosine fp32" "8192 16384 256 L2 bf16" "3072 3072 256 Cosine fp32" "6272 8192 768 Cosine fp32" "65536 16384 256 L2 bf16"; do
  python tools/phase_stamps.py $sh 2>&1 | grep -v "amdgpu.ids\|Warning"
done > gpurun_out/r05_phase_stamps.txt 2>&1
unset VQHIP_LIB
VQ_EXP_ENCODE=1 python tools/exp_shape.py 524288 16384 8 L2 shipped build/exp/libvqhip_w32occ3.so > gpurun_out/r05_w32occ3.txt 2>&1
python tools/bench_train_shapes.py > gpurun_out/r05_train_shapes_again.txt 2>&1
