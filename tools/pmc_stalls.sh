#!/bin/bash
# Stall-attribution PMC passes for one encode shape (run on the GPU box).  usage: tools/pmc_stalls.sh <outdir-under-gpurun_out> N K D L2|Cosine
out=/root/repo/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
SHAPE="$*"
run() { name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 /root/repo/tools/prof_shape.py $SHAPE 6 > $out/$name.log 2>&1; echo "$name rc=$?"; }
run a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC
run b SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH SQ_IFETCH_LEVEL
run c SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_BRANCH SQ_BUSY_CYCLES SQ_CYCLES
run d SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES
python3 - $out "$SHAPE" <<'PY'
import csv, glob, sys, json, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}
res['_shape'] = sys.argv[2]
json.dump(res, open(out + '/pmc_stalls.json', 'w'), indent=1)
for k, d in res.items():
    if 'coarse' in k or 'exact_tiled' in k:
        print(k)
        for c in sorted(d): print(f'   {c:32s} {d[c]:16.1f}')
PY
