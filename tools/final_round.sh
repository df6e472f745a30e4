#!/bin/bash
# The measurement set of a round with the library that is in the tree (run on the GPU box): tools/final_round.sh <tag, e.g. r04>
# -> gpurun_out/<tag>_*.  Copy what is to be judged into profiles/.
tag=$1
cd /root/repo
o=gpurun_out
sha256sum vector_quantization_amd/libvqhip.so > $o/${tag}_lib_sha256.txt
python bench.py > $o/${tag}_bench.json 2> $o/${tag}_bench.err; echo "bench rc=$?"
bash tools/prof_bench.sh ${tag}_bench_profiled --no-cpu-baseline --min-seconds 0 > $o/${tag}_bench_profiled.txt 2>&1; tail -3 $o/${tag}_bench_profiled.txt
bash tools/pmc_passes.sh ${tag}_pmc > $o/${tag}_pmc.txt 2>&1; tail -3 $o/${tag}_pmc.txt | cut -c1-300
python bench.py --workload cvq --no-cpu-baseline --min-seconds 3 > $o/${tag}_cvq.json 2>> $o/${tag}_bench.err; echo "cvq rc=$?"
python bench.py --workload cvq --images 256 --no-cpu-baseline --min-seconds 3 > $o/${tag}_cvq256.json 2>> $o/${tag}_bench.err; echo "cvq256 rc=$?"
python bench.py --workload vqkd --no-cpu-baseline --min-seconds 3 > $o/${tag}_vqkd.json 2>> $o/${tag}_bench.err; echo "vqkd rc=$?"
python bench.py --workload tokenize --no-cpu-baseline > $o/${tag}_tokenize.json 2>> $o/${tag}_bench.err; echo "tokenize rc=$?"
python bench.py --images 32 --no-cpu-baseline --min-seconds 3 > $o/${tag}_bench_32img.json 2>> $o/${tag}_bench.err; echo "32img rc=$?"
python bench.py --images 256 --no-cpu-baseline --min-seconds 3 > $o/${tag}_bench_256img.json 2>> $o/${tag}_bench.err; echo "256img rc=$?"
VQ_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 --images 256 --min-seconds 0.3 --no-cpu-baseline > $o/${tag}_gpus2_shared.json 2>> $o/${tag}_bench.err; echo "gpus2 rc=$?"
python tools/bench_shapes.py > $o/${tag}_shapes.txt 2>&1; echo "shapes rc=$?"
python tools/time_exact_tiled.py >> $o/${tag}_shapes.txt 2>&1
# BASELINE configs[2] / configs[4]: kernel statistics and counters of the D <= 32 kernels on the final library
bash tools/prof_shape.sh ${tag}_c3_prof 100352 8192 32 Cosine > $o/${tag}_c3_prof.txt 2>&1
bash tools/pmc_shape.sh ${tag}_c3_pmc 100352 8192 32 Cosine > $o/${tag}_c3_pmc.txt 2>&1
bash tools/pmc_stalls.sh ${tag}_c3_stalls 100352 8192 32 Cosine > $o/${tag}_c3_stalls.txt 2>&1
bash tools/prof_shape.sh ${tag}_tok_prof 524288 16384 8 Cosine > $o/${tag}_tok_prof.txt 2>&1
bash tools/pmc_shape.sh ${tag}_tok_pmc 524288 16384 8 Cosine > $o/${tag}_tok_pmc.txt 2>&1
# per-launch timelines of the one-call encode at the small shapes
(VQ_PROF_ENCODE=1 VQ_PROF_BF16=1 bash tools/timeline_shape.sh ${tag}_tl_c2_32img pre_kernel 8192 16384 256 L2; VQ_PROF_ENCODE=1 bash tools/timeline_shape.sh ${tag}_tl_c3 pre_kernel 100352 8192 32 Cosine; VQ_PROF_ENCODE=1 bash tools/timeline_shape.sh ${tag}_tl_c4 pre_kernel 3072 16384 256 Cosine) > $o/${tag}_timelines.txt 2>&1
# RCCL at world size 1: the test, then the A/B of the two all-reduce routes
timeout 900 python -m pytest tests/test_gpu_rccl.py -q > $o/${tag}_rccl_test.log 2>&1; echo "rccl test rc=$?"
bash tools/rccl_ws1_ab.sh ${tag}_rccl_ab 3 > $o/${tag}_rccl_ab.txt 2>&1; cat $o/${tag}_rccl_ab.txt
# round 5: training steps at the per-rank shapes of the shipped configs (eager one-call and graph-replayed) and their per-launch timelines
python tools/bench_train_shapes.py cvq vqkd cluster llamagen cvq64k 2>&1 | grep "ms per step\|bound to\|listed for" > $o/${tag}_train_shapes.txt; cat $o/${tag}_train_shapes.txt
(bash tools/timeline_train.sh ${tag}_tl_cvq cvq vq_backward_kernel; bash tools/timeline_train.sh ${tag}_tl_vqkd vqkd vqkd_backward_kernel; bash tools/timeline_train.sh ${tag}_tl_cluster cluster vq_backward_kernel; bash tools/timeline_train.sh ${tag}_tl_llamagen llamagen normalize_bwd_kernel) > $o/${tag}_train_timelines.txt 2>&1
# round 6: the closing randomised campaigns (pipeline against the all-fp32 route — whose whole-batch pass is exact_stream_kernel now — and, last line,
# the streamed against the register form of that pass itself under tools/time_exact_tiled.py's shapes)
{ sha256sum vector_quantization_amd/libvqhip.so; python tools/fuzz_vs_exact.py 150 2>&1 | tail -1; VQ_FUZZ_BF16=1 VQ_FUZZ_DIMS=256 python tools/fuzz_vs_exact.py 100 2>&1 | tail -1; VQ_FUZZ_DIMS=8,16,24,32 python tools/fuzz_vs_exact.py 100 2>&1 | tail -1; VQ_FUZZ_DIMS=512,768,1024 python tools/fuzz_vs_exact.py 60 2>&1 | tail -1; VQ_FUZZ_FORCE_EXACT=1 python tools/fuzz_vs_exact.py 60 2>&1 | tail -1; } > $o/${tag}_final_fuzz.txt 2>&1; cat $o/${tag}_final_fuzz.txt
