cd /root/repo
bash tools/prof_shape.sh r04_c3_prof_b 100352 8192 32 Cosine
bash tools/prof_shape.sh r04_tok_prof_b 524288 16384 8 Cosine
