#!/bin/bash
# The measurement set of a round with the library that is in the tree (run on the GPU box): tools/final_round.sh <tag, e.g. r03>
# -> gpurun_out/<tag>_*.  Copy what is to be judged into profiles/.
tag=$1
cd /root/repo
o=gpurun_out
python bench.py > $o/${tag}_bench.json 2> $o/${tag}_bench.err; echo "bench rc=$?"
bash tools/prof_bench.sh ${tag}_bench_profiled --no-cpu-baseline --min-seconds 0 > $o/${tag}_bench_profiled.txt 2>&1; tail -3 $o/${tag}_bench_profiled.txt
bash tools/pmc_passes.sh ${tag}_pmc > $o/${tag}_pmc.txt 2>&1; tail -3 $o/${tag}_pmc.txt
python bench.py --workload cvq > $o/${tag}_cvq.json 2>> $o/${tag}_bench.err; echo "cvq rc=$?"
python bench.py --workload cvq --images 256 > $o/${tag}_cvq256.json 2>> $o/${tag}_bench.err; echo "cvq256 rc=$?"
python bench.py --workload tokenize > $o/${tag}_tokenize.json 2>> $o/${tag}_bench.err; echo "tokenize rc=$?"
python bench.py --images 32 --no-cpu-baseline > $o/${tag}_bench_32img.json 2>> $o/${tag}_bench.err; echo "32img rc=$?"
python bench.py --images 256 --no-cpu-baseline > $o/${tag}_bench_256img.json 2>> $o/${tag}_bench.err; echo "256img rc=$?"
VQ_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 10 --warmup 3 --images 256 --min-seconds 0.3 --no-cpu-baseline > $o/${tag}_gpus2_shared.json 2>> $o/${tag}_bench.err; echo "gpus2 rc=$?"
python tools/bench_shapes.py > $o/${tag}_shapes.txt 2>&1; echo "shapes rc=$?"
python tools/time_exact_tiled.py >> $o/${tag}_shapes.txt 2>&1
