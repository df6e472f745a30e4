cd /root/repo
o=gpurun_out
sha256sum vector_quantization_amd/libvqhip.so > $o/r04_lib_sha.txt
timeout 900 python -m pytest tests/test_gpu_rccl.py -x -q > $o/r04_rccl_test.log 2>&1; echo "rccl test rc=$?"; tail -5 $o/r04_rccl_test.log
for route in torch direct; do
  VQ_FORCE_EXCHANGE=1 VQHIP_ALLREDUCE=$route timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29511 bench.py --workload cvq --min-seconds 2 > $o/r04_rccl_ws1_$route.json 2> $o/r04_rccl_ws1_$route.err; echo "bench $route rc=$?"
done
timeout 300 python bench.py --workload cvq --min-seconds 2 > $o/r04_cvq_nogroup.json 2> $o/r04_cvq_nogroup.err; echo "nogroup rc=$?"
bash tools/prof_shape.sh r04_c3_prof 100352 8192 32 Cosine
bash tools/pmc_shape.sh r04_c3_pmc 100352 8192 32 Cosine
bash tools/pmc_stalls.sh r04_c3_stalls 100352 8192 32 Cosine
bash tools/prof_shape.sh r04_tok_prof 524288 16384 8 Cosine
bash tools/pmc_shape.sh r04_tok_pmc 524288 16384 8 Cosine
