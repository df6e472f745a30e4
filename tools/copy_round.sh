#!/bin/bash
# After tools/final_round.sh <tag> has run on the GPU box: copy what is to be judged from gpurun_out/ into profiles/ and make
# profiles/pmc_latest.json the PMC summary of this library.  usage: tools/copy_round.sh <tag> "<note for pmc_latest.json>"
set -e
tag=$1; note=$2
cd "$(dirname "${BASH_SOURCE[0]}")/.."
o=gpurun_out; p=profiles
latest() { ls -t $1 2>/dev/null | head -1; }
for f in bench bench_profiled bench_32img bench_256img cvq cvq256 vqkd tokenize gpus2_shared; do cp $o/${tag}_$f.json $p/${tag}_$f.json; done
cp $o/${tag}_lib_sha256.txt $o/${tag}_shapes.txt $o/${tag}_timelines.txt $o/${tag}_rccl_ab.txt $o/${tag}_train_shapes.txt $o/${tag}_train_timelines.txt $p/
cp "$(latest "$o/${tag}_bench_profiled/*/*kernel_stats.csv")" $p/${tag}_kernel_stats.csv
cp "$(latest "$o/${tag}_c3_prof/*/*kernel_stats.csv")" $p/${tag}_c3_kernel_stats.csv
cp "$(latest "$o/${tag}_tok_prof/*/*kernel_stats.csv")" $p/${tag}_tok_kernel_stats.csv
cp $o/${tag}_pmc/pmc_summary.json $p/${tag}_pmc_summary.json
cp $o/${tag}_c3_pmc/pmc_summary.json $p/${tag}_c3_pmc_summary.json
cp $o/${tag}_c3_stalls/pmc_stalls.json $p/${tag}_c3_pmc_stalls.json
cp $o/${tag}_tok_pmc/pmc_summary.json $p/${tag}_tok_pmc_summary.json
python tools/pmc_to_latest.py $o/${tag}_pmc/pmc_summary.json 524288 "$note"
cat $p/${tag}_lib_sha256.txt
cp $o/${tag}_final_fuzz.txt $p/
