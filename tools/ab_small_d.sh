#!/bin/bash
# A/B of libvqhip builds over the D <= 32 shapes (run on the GPU box): tools/ab_small_d.sh lib1 lib2 ...   ('shipped' = in-tree)
# one-call encodes (the training-time form), alternating subprocess rounds per shape (tools/exp_shape.py)
for shape in "4096 8192 32 Cosine" "8192 8192 32 Cosine" "12544 8192 32 Cosine" "16384 8192 32 Cosine" "32768 8192 32 Cosine" "65536 8192 32 Cosine" "100352 8192 32 Cosine" \
             "4096 16384 8 L2" "8192 16384 8 L2" "16384 16384 8 L2" "65536 16384 8 L2" "524288 16384 8 L2" "12544 8192 32 L2" "65536 8192 16 Cosine"; do
    VQ_EXP_ENCODE=1 python3 tools/exp_shape.py $shape "$@" 2>&1 | tail -$(( $# + 1 ))
done
